// save_png_test.cpp -- RGB8Image::save of the C++ mirror writes a PNG any reader accepts (tests/test_host_loader_cpu.py decodes it)
#include "../../jtx-pathtracer_amd/host/jtx_host_api.hpp"
int main(int argc, char **argv) {
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    jtxmi::RGB8Image img(w, h);
    for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) img.data()[(size_t) r * w + c] = jtxmi::RGB{(unsigned char) (r * 7 + c), (unsigned char) (c * 3), (unsigned char) (r ^ c)};
    img.save(argv[1]);
    return 0;
}
