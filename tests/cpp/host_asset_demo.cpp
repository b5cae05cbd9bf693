// host_asset_demo.cpp -- the C++ host end to end: loadScene (glTF / GLB or OBJ, host/jtx_host_loader.hpp) -> buildBVH ->
// StaticCamera::render on the GPU -> Camera::save as PNG.  Prints the CRC-32 of the RGB8 image for the test to compare with the
// same asset rendered through the Python mirror (the Cornell-room camera and point light of scenes.mixed()).
#include "../../jtx-pathtracer_amd/host/jtx_host_loader.hpp"
#include <cstdio>
using namespace jtxmi;
int main(int argc, char **argv) {
    try {
        Scene scene; scene.name = "File scene";
        loadScene(argv[1], scene);
        scene.skyColor = Vec3(0.5f, 0.7f, 1.0f);
        Light l; l.type = Light::POINT; l.position = Vec3(278, 500, 279.5f); l.intensity = Vec3(1, 1, 1); l.scale = 60000;
        scene.lights.push_back(l);
        scene.cameraProperties.center = Vec3(278, 273, -800); scene.cameraProperties.target = Vec3(278, 273, 0); scene.cameraProperties.up = Vec3(0, 1, 0);
        scene.cameraProperties.yfov = 39.3077f; scene.cameraProperties.defocusAngle = 0; scene.cameraProperties.focusDistance = 1;
        scene.buildBVH();
        StaticCamera camera(96, 72, scene.cameraProperties, 2, 2, 5);
        camera.renderFinal(scene);
        unsigned c = 0xffffffffu; const unsigned char *b = &camera.img_.data()[0].R;
        for (int i = 0; i < 96 * 72 * 3; ++i) { c ^= b[i]; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u))); }
        std::printf("meshes %zu triangles %zu textures %zu crc %08x\n", scene.meshes.size(), scene.triangles.size(), scene.textures.size(), ~c);
        if (argc > 2) camera.save(argv[2]);
        return 0;
    } catch (const std::exception &e) { std::printf("error %s\n", e.what()); return 1; }
}
