// tools/fc_ubench.hip — stand-alone timing of the FC-heads kernels (cz_heads.hip).  args: B iters
// Build variants with -DCZ_PFC_NOSTORE (no logits stores) to separate compute from the output write.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cchess_zero_amd/csrc/cz_heads.hip"
void cz_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192, iters = argc > 2 ? atoi(argv[2]) : 20;
    float *z, *pb, *v1w, *v1b, *v2w, *v2b, *logits, *value; void *whi, *wlo;
    CK(hipMalloc(&z, (size_t)B * 270 * 4)); CK(hipMalloc(&logits, (size_t)B * 2086 * 4)); CK(hipMalloc(&value, (size_t)B * 4));
    CK(hipMalloc(&whi, 66 * 12 * 64 * 16)); CK(hipMalloc(&wlo, 66 * 12 * 64 * 16)); CK(hipMalloc(&pb, 2086 * 4));
    CK(hipMalloc(&v1w, 90 * 256 * 4)); CK(hipMalloc(&v1b, 1024)); CK(hipMalloc(&v2w, 1024)); CK(hipMalloc(&v2b, 4));
    std::vector<float> hz((size_t)B * 270);
    srand(1);
    for (auto &x : hz) x = (rand() & 1) ? (rand() % 1000) * 1e-3f : 0.f;
    CK(hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(whi, 0x3c, 66 * 12 * 64 * 16)); CK(hipMemset(wlo, 0x30, 66 * 12 * 64 * 16)); CK(hipMemset(pb, 0, 2086 * 4));
    CK(hipMemset(v1w, 0, 90 * 256 * 4)); CK(hipMemset(v1b, 0, 1024)); CK(hipMemset(v2w, 0, 1024)); CK(hipMemset(v2b, 0, 4));
    cz_ctx c{}; c.device = 0; c.stream = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i)
                cz_fc_heads_f32(&c, z, whi, wlo, pb, v1w, v1b, v2w, v2b, which == 0 ? logits : nullptr, which == 1 ? value : nullptr, B);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%s B=%d: %.1f us/call\n", which == 0 ? "policy_fc" : "value_fc", B, ms * 1e3 / iters);
        }
    }
    return 0;
}
