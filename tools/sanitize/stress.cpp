// Stress driver of the sanitizer builds (`make sanitize`): HostPool, the seam's lock order across two caller threads and a fork, the banded add, and the
// run-time instantiations' code cache from several threads at once.  Built twice, with -fsanitize=thread and -fsanitize=address,undefined (Makefile).
// usage: stress_<san> [iterations]      exit code 0 = every check passed (the sanitizer itself aborts or reports on stderr otherwise)
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <initializer_list>

extern "C" {
int kyhostcheck_seam_stress(int iterations);
int kyhostcheck_add_rows(int width, int height, int stride_px, int n_threads, int rounds);
int kyhostcheck_chunks(int spp);
int kyhostcheck_jit_stress(int n_threads, int rounds);
const char* kyhip_jit_status(void);
}

int main(int argc, char** argv) {
    const int it = argc > 1 ? std::atoi(argv[1]) : 2000;
    int rc = kyhostcheck_seam_stress(it);
    std::printf("seam_stress(%d) -> %d\n", it, rc);
    if (rc) return 1;
    rc = kyhostcheck_add_rows(253, 97, 260, 4, 5);
    std::printf("add_rows -> %d\n", rc);
    if (rc) return 2;
    for (int spp : {1, 2, 3, 4, 5, 16, 63, 64, 65, 447, 448, 449, 472, 1024, 4096, 16384, 100003})
        if (kyhostcheck_chunks(spp) < 1) { std::printf("chunks(%d) failed\n", spp); return 3; }
    std::printf("chunk schedules ok\n");
    if (std::getenv("KYHIP_HIPCC")) {   // the cache's threads: only with a stand-in compiler (the real one takes seconds per object)
        const int got = kyhostcheck_jit_stress(6, 4);
        std::printf("jit_stress -> %d objects of 24 requests; status: %s\n", got, kyhip_jit_status());
        if (got != 24) return 4;
    }
    return 0;
}
