// ref_serial_driver.cc -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, never shipped).
//
// A command-line driver around the REFERENCE's own serial tracer headers, compiled from
// where they lie (-I/root/reference/src/InOneWeekend; see oracle/Makefile).  The reference's
// main.cc hard-codes scene 1 / 1280x768 / 10 spp / depth 20 (src/InOneWeekend/main.cc:69-73)
// and has no CLI; this driver only supplies the parameters and the scene-id grid ranges of
// src/GlobalFloatCUDAInOneWeekend/main.cu:148-284.  Every class used (camera, sphere,
// hittable_list, lambertian, metal, dielectric, vec3, ...) is the reference's.
//
//   ref_serial_driver <scene_id> <width> <height> <samples> <depth>              -> P3 on stdout
//   ref_serial_driver <scene_id> <width> <height> <samples> <depth> <row_step>   -> the rows row_step/2, row_step/2 + row_step, ... of the SAME
//       view only (bench.py's bounded CPU sample of the benchmark frame itself): the reference's own initialize / get_ray / ray_color /
//       write_color called row by row -- those members are private to its camera class, hence the access define below; the header reads
//       "P3 / width rows / 255".  The random stream of a row subset is not the full render's, which a throughput sample does not need.
#include "rtweekend.h"
#define private public
#include "camera.h"
#undef private
#include "hittable.h"
#include "hittable_list.h"
#include "material.h"
#include "sphere.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) { std::fprintf(stderr, "usage: %s scene_id width height samples depth [row_step]\n", argv[0]); return 2; }
    const int row_step = argc == 7 ? std::atoi(argv[6]) : 0;
    const int scene_id = std::atoi(argv[1]), W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    const int S = std::atoi(argv[4]), depth = std::atoi(argv[5]);
    int a0, a1, b0, b1;
    if (scene_id == 1) { a0 = -11; a1 = 11; b0 = -11; b1 = 11; }
    else if (scene_id == 2) { a0 = 5; a1 = 11; b0 = 5; b1 = 11; }
    else { a0 = -11; a1 = 0; b0 = -11; b1 = 0; }

    hittable_list world;
    world.add(make_shared<sphere>(point3(0,-1000,0), 1000, make_shared<lambertian>(color(0.5, 0.5, 0.5))));
    for (int a = a0; a < a1; a++) {
        for (int b = b0; b < b1; b++) {
            auto choose_mat = random_double();
            point3 center(a + 0.9*random_double(), 0.2, b + 0.9*random_double());
            if ((center - point3(4, 0.2, 0)).length() > 0.9) {
                shared_ptr<material> m;
                if (choose_mat < 0.8) {
                    auto albedo = color::random() * color::random();
                    m = make_shared<lambertian>(albedo);
                } else if (choose_mat < 0.95) {
                    auto albedo = color::random(0.5, 1);
                    auto fuzz = random_double(0, 0.5);
                    m = make_shared<metal>(albedo, fuzz);
                } else {
                    m = make_shared<dielectric>(1.5);
                }
                world.add(make_shared<sphere>(center, 0.2, m));
            }
        }
    }
    world.add(make_shared<sphere>(point3(0, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    world.add(make_shared<sphere>(point3(-4, 1, 0), 1.0, make_shared<lambertian>(color(0.4, 0.2, 0.1))));
    world.add(make_shared<sphere>(point3(4, 1, 0), 1.0, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));

    camera cam;
    cam.aspect_ratio      = double(W) / double(H);
    cam.image_width       = W;
    cam.samples_per_pixel = S;
    cam.max_depth         = depth;
    cam.vfov     = 20;
    cam.lookfrom = point3(13,2,3);
    cam.lookat   = point3(0,0,0);
    cam.vup      = vec3(0,1,0);
    cam.defocus_angle = 0.6;
    cam.focus_dist    = 10.0;
    // camera::initialize derives the height as int(width / aspect_ratio)
    // (src/InOneWeekend/camera.h:69); refuse sizes where that does not give back H.
    if (int(W / cam.aspect_ratio) != H) { std::fprintf(stderr, "height %d not representable via aspect ratio\n", H); return 3; }
    if (row_step <= 0) { cam.render(world); return 0; }
    cam.initialize();
    int rows = 0;
    for (int j = row_step / 2; j < cam.image_height; j += row_step) ++rows;
    std::cout << "P3\n" << cam.image_width << ' ' << rows << "\n255\n";
    for (int j = row_step / 2; j < cam.image_height; j += row_step)
        for (int i = 0; i < cam.image_width; i++) {
            color pixel_color(0, 0, 0);
            for (int sample = 0; sample < cam.samples_per_pixel; sample++) pixel_color += cam.ray_color(cam.get_ray(i, j), cam.max_depth, world, j, i);
            write_color(std::cout, cam.pixel_samples_scale * pixel_color);
        }
    return 0;
}
