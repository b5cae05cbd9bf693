// host_api_demo.cpp -- drives the C++ host mirror the way the reference's main.cpp / Display::renderScene do:
// createMeshScene (scene.cpp:137-174, with UVs and tex ids = -1) -> buildBVH -> StaticCamera{64,64,...,2,2,4} -> render.
// Prints the FNV-1a hash of the RGB8 image; the reference's own sources give 1af9ba89 for this set-up.
#include "../../jtx-pathtracer_amd/host/jtx_host_api.hpp"
#include <cstdio>
using namespace jtxmi;

int main() {
    try {
        Scene scene; scene.name = "Mesh Scene";
        scene.cameraProperties.center = Vec3(0, 0, 8); scene.cameraProperties.target = Vec3(0, 0, -1); scene.cameraProperties.up = Vec3(0, 1, 0);
        scene.cameraProperties.yfov = 20; scene.cameraProperties.defocusAngle = 0; scene.cameraProperties.focusDistance = 3.4f;
        scene.skyColor = Vec3(0.7f, 0.8f, 1.0f);
        scene.materials.reserve(8);
        Material m; m.type = Material::DIFFUSE; m.albedo = Vec3(1, 0.3f, 0.5f);
        scene.materials.push_back(m);
        Vec3 vertices[4] = {Vec3(-1, -1, -1), Vec3(-1, 1, -1), Vec3(1, 1, -1), Vec3(1, -1, -1)};
        Vec3i indices[2] = {Vec3i(0, 1, 2), Vec3i(0, 2, 3)};
        Vec3 normals[4] = {Vec3(0, 0, 1), Vec3(0, 0, 1), Vec3(0, 0, 1), Vec3(0, 0, 1)};
        Vec2f uvs[4] = {{0, 0}, {0, 1}, {1, 1}, {1, 0}};
        scene.meshes.push_back(Mesh(indices, 2, vertices, 4, normals, uvs, &scene.materials.back()));
        scene.triangles.push_back({0, 0}); scene.triangles.push_back({1, 0});
        scene.buildBVH();
        StaticCamera camera(64, 64, scene.cameraProperties, 2, 2, 4);
        camera.render(scene);                       // progressive, one pass per stratum
        unsigned h = 2166136261u;
        const unsigned char *b = &camera.img_.data()[0].R;
        for (int i = 0; i < 64 * 64 * 3; ++i) { h ^= b[i]; h *= 16777619u; }
        // single-ray API: the centre ray must hit the quad at t = 9 (camera z = 8, quad z = -1)
        SurfaceIntersection rec;
        const bool hit = scene.closestHit(Ray(Vec3(0.1f, 0.2f, 8), Vec3(0, 0, -1)), Interval(0.001f, INF), rec);
        const bool shadow = scene.anyHit(Ray(Vec3(0.1f, 0.2f, 8), Vec3(0, 0, -1)), Interval(0.0f, 5.0f));
        const AABB box = scene.bounds();            // scene.hpp:71-74
        std::printf("hash %08x samples %d hit %d t %.3f shadow %d radius %.4f boundsmin %g,%g,%g boundsmax %g,%g,%g\n", h, camera.currentSample_.load(),
                    (int) hit, rec.t, (int) shadow, scene.getSceneRadius(), box.pmin.x, box.pmin.y, box.pmin.z, box.pmax.x, box.pmax.y, box.pmax.z);
        // the edit loop without the host (display.cpp:902-905): Scene::rebuildBVH on the device, same frame
        scene.rebuildBVHOnDevice();
        StaticCamera again(64, 64, scene.cameraProperties, 2, 2, 4);
        again.renderFinal(scene);
        unsigned hr = 2166136261u;
        const unsigned char *br = &again.img_.data()[0].R;
        for (int i = 0; i < 64 * 64 * 3; ++i) { hr ^= br[i]; hr *= 16777619u; }
        std::printf("rebuildhash %08x\n", hr);
        // DynamicCamera (the UI's progressive camera): non-blocking render, restart on a camera change
        DynamicCamera dyn(64, 64, scene.cameraProperties, 2, 2, 4, 1);
        dyn.render(scene); dyn.wait();
        unsigned hd = 2166136261u;
        const unsigned char *bd = &dyn.img_.data()[0].R;
        for (int i = 0; i < 64 * 64 * 3; ++i) { hd ^= bd[i]; hd *= 16777619u; }
        dyn.properties_.center = Vec3(0.5f, 0.25f, 8);                  // move the camera and start over, twice in a row
        dyn.render(scene); dyn.render(scene); dyn.wait();
        StaticCamera moved(64, 64, dyn.properties_, 2, 2, 4);
        moved.renderFinal(scene);
        int same = 1;
        for (int i = 0; i < 64 * 64; ++i) {
            const RGB a = dyn.img_.data()[i], c = moved.img_.data()[i];
            if (a.R != c.R || a.G != c.G || a.B != c.B) same = 0;
        }
        dyn.stopRender();
        std::printf("dynhash %08x dynsamples %d restart_same %d\n", hd, dyn.currentSample_.load(), same);
        // the same frame over 3 shards through jtx_mi_multi_render (Scene::useDevices; all on device 0 here, 8 GPUs on a node)
        scene.destroy();
        scene.useDevices({0, 0, 0});
        scene.buildBVH();
        StaticCamera mcam(64, 64, scene.cameraProperties, 2, 2, 4);
        mcam.render(scene);
        unsigned hm = 2166136261u;
        const unsigned char *bm = &mcam.img_.data()[0].R;
        for (int i = 0; i < 64 * 64 * 3; ++i) { hm ^= bm[i]; hm *= 16777619u; }
        std::printf("multihash %08x multisamples %d\n", hm, mcam.currentSample_.load());
        scene.destroy();
        return 0;
    } catch (const std::exception &e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
}
