// smallpt's main() (smallpt2pbrt/smallpt.cpp:91-123) on the GPU: the loop nest is kyhip_smallpt_render, the output code --
// gamma 2.2, 8 bits, plain-text PPM "image.ppm" -- is smallpt's own (54-55, 119-123).
//   usage: smallpt_driver [spp = 40] [width = 1024] [height = 768] [output = image.ppm]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/kyhip.h"

static double clamp01(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; }
static int to_int(double x) { return int(std::pow(clamp01(x), 1 / 2.2) * 255 + .5); }   // toInt, 55

int main(int argc, char* argv[]) {
    const int samps = (argc > 1 ? std::atoi(argv[1]) : 40) / 4;   // smallpt.cpp:92
    const int w = argc > 2 ? std::atoi(argv[2]) : 1024, h = argc > 3 ? std::atoi(argv[3]) : 768;
    const char* path = argc > 4 ? argv[4] : "image.ppm";
    ky_smallpt_sphere spheres[9];
    const int n = kyhip_smallpt_scene(spheres);
    ky_smallpt_params p = {w, h, samps < 1 ? 1 : samps, 1234u, 10, KY_SP_VARIANT_SMALLPT};
    std::vector<double> c(3 * (size_t)w * h);
    if (kyhip_smallpt_render(0, spheres, n, &p, c.data()) != KY_OK) {
        std::fprintf(stderr, "error: %s\n", kyhip_last_error());
        return 1;
    }
    std::FILE* f = std::fopen(path, "w");
    if (!f) { std::perror(path); return 1; }
    std::fprintf(f, "P3\n%d %d\n%d\n", w, h, 255);
    for (int i = 0; i < w * h; i++) std::fprintf(f, "%d %d %d ", to_int(c[3 * i]), to_int(c[3 * i + 1]), to_int(c[3 * i + 2]));
    std::fclose(f);
    std::fprintf(stderr, "%s: %dx%d, %d spp, kernel %.2f ms\n", path, w, h, 4 * p.samps, kyhip_kernel_ms(0));
    return 0;
}
