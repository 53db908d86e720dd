// cz_conv.hip — C-ABI wrapper of the fused MFMA conv3x3 kernel (device code: cz_conv_kernel.h).
#include "cz_internal.h"
#include "cz_conv_kernel.h"
#include "cz_trunk_split.h"
#include "cz_trunk_mx.h"
#ifdef CZ_EXPERIMENT_MX2   /* tools/experiments/mx_ablate.sh: the 3 x 2-tile / K-split variant of round 6 (measured, not adopted: DESIGN.md 4.2) */
#include "cz_trunk_mx2.h"
#include <cstdlib>
#endif
#ifdef CZ_EXPERIMENT_MX12  /* the same arithmetic with twelve waves per workgroup, two cell tiles per wave */
#include "cz_trunk_mx12.h"
#include <cstdlib>
#endif

extern "C" int cz_conv3x3_c128_bf16(cz_ctx *c, const void *in, const void *wpk, const float *bias, const void *residual,
                                    void *out, int B, int relu) {
    using namespace czconv;
    CZ_REQUIRE(c && in && wpk && bias && out && B >= 0, "cz_conv3x3_c128_bf16: null argument");
    if (B == 0) return CZ_OK;
    if (!c->conv_attr_set) {
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_c128), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES));
        c->conv_attr_set = true;
    }
    const int grid = (B + CV_P - 1) / CV_P;
    hipLaunchKernelGGL(k_conv3x3_c128, dim3(grid), dim3(CV_THREADS), CV_LDS_BYTES, c->stream, (const uint16_t *)in,
                       (const uint16_t *)wpk, bias, (const uint16_t *)residual, (uint16_t *)out, B, relu);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

// cz_set_clock_probe: the trunk kernels stamp every workgroup's start / end in both clocks when the buffer holds the grid
static unsigned long long *clock_probe(cz_ctx *c, int grid) {
    if (!c->clock_probe || grid > c->clock_probe_wgs) return nullptr;
    c->clock_probe_last_grid = grid;
    return c->clock_probe;
}

static int launch_tower(cz_ctx *c, const void *in, const void *wpk, const float *bias, void *out, const float *head_w,
                        const float *head_b, float *head_out, int B, int nblocks, const void *planes = nullptr,
                        const void *w0 = nullptr, const float *b0 = nullptr, bool f16 = false) {
    if (head_w && (reinterpret_cast<uintptr_t>(head_w) & 15u)) { cz_set_error("cz_net_trunk: head_w must be 16-byte aligned"); return CZ_EINVAL; }
    using namespace czconv;
    if (B == 0) return CZ_OK;
    if (!c->tower_attr_set) {
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
        c->tower_attr_set = true;
    }
    const int grid = (B + T8_P - 1) / T8_P;
    if (f16)
        hipLaunchKernelGGL((k_tower8_c128<true, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, c->stream, (const uint16_t *)in,
                           (const uint16_t *)wpk, bias, (uint16_t *)out, head_w, head_b, head_out, (const uint16_t *)planes,
                           (const uint16_t *)w0, b0, B, 2 * nblocks, c->batch_count, clock_probe(c, grid));
    else
        hipLaunchKernelGGL((k_tower8_c128<false, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, c->stream, (const uint16_t *)in,
                           (const uint16_t *)wpk, bias, (uint16_t *)out, head_w, head_b, head_out, (const uint16_t *)planes,
                           (const uint16_t *)w0, b0, B, 2 * nblocks, c->batch_count, clock_probe(c, grid));
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

extern "C" int cz_tower_c128_bf16(cz_ctx *c, const void *in, const void *wpk, const float *bias, void *out, int B, int nblocks) {
    CZ_REQUIRE(c && in && wpk && bias && out && B >= 0 && nblocks >= 1, "cz_tower_c128_bf16: null argument / nblocks < 1");
    return launch_tower(c, in, wpk, bias, out, nullptr, nullptr, nullptr, B, nblocks);
}

extern "C" int cz_tower_heads_c128_bf16(cz_ctx *c, const void *in, const void *wpk, const float *bias, void *trunk_out,
                                        const float *head_w, const float *head_b, float *head_out, int B, int nblocks) {
    CZ_REQUIRE(c && in && wpk && bias && head_w && head_b && head_out && B >= 0 && nblocks >= 1,
               "cz_tower_heads_c128_bf16: null argument / nblocks < 1");
    return launch_tower(c, in, wpk, bias, trunk_out, head_w, head_b, head_out, B, nblocks);
}

extern "C" int cz_net_trunk_bf16(cz_ctx *c, const void *planes16, const void *w0, const float *b0, const void *wpk,
                                 const float *bias, void *trunk_out, const float *head_w, const float *head_b,
                                 float *head_out, int B, int nblocks) {
    CZ_REQUIRE(c && planes16 && w0 && b0 && wpk && bias && B >= 0 && nblocks >= 1 && (trunk_out || head_out),
               "cz_net_trunk_bf16: null argument / nblocks < 1");
    CZ_REQUIRE(!head_out || (head_w && head_b), "cz_net_trunk_bf16: head_out needs head_w and head_b");
    return launch_tower(c, nullptr, wpk, bias, trunk_out, head_w, head_b, head_out, B, nblocks, planes16, w0, b0);
}

extern "C" int cz_net_trunk_f16(cz_ctx *c, const void *planes16, const void *w0, const float *b0, const void *wpk,
                                const float *bias, void *trunk_out, const float *head_w, const float *head_b,
                                float *head_out, int B, int nblocks) {
    CZ_REQUIRE(c && planes16 && w0 && b0 && wpk && bias && B >= 0 && nblocks >= 1 && (trunk_out || head_out),
               "cz_net_trunk_f16: null argument / nblocks < 1");
    CZ_REQUIRE(!head_out || (head_w && head_b), "cz_net_trunk_f16: head_out needs head_w and head_b");
    return launch_tower(c, nullptr, wpk, bias, trunk_out, head_w, head_b, head_out, B, nblocks, planes16, w0, b0, true);
}

extern "C" int cz_net_trunk_split(cz_ctx *c, const void *planes16, const void *w0, const float *b0, const void *wpk,
                                  const float *bias, float *trunk_out, const float *head_w, const float *head_b,
                                  float *head_out, int B, int nblocks, int halves_dtype) {
    using namespace czconv;
    CZ_REQUIRE(c && planes16 && w0 && b0 && wpk && bias && B >= 0 && nblocks >= 1 && (trunk_out || head_out),
               "cz_net_trunk_split: null argument / nblocks < 1");
    CZ_REQUIRE(!head_out || (head_w && head_b), "cz_net_trunk_split: head_out needs head_w and head_b");
    CZ_REQUIRE(halves_dtype == CZ_F16 || halves_dtype == CZ_BF16, "cz_net_trunk_split: halves_dtype must be CZ_F16 or CZ_BF16");
    if (head_w && (reinterpret_cast<uintptr_t>(head_w) & 15u)) { cz_set_error("cz_net_trunk_split: head_w must be 16-byte aligned"); return CZ_EINVAL; }
    if (B == 0) return CZ_OK;
    if (!c->split_attr_set) {
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trunk_split_c128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, XS_LDS_BYTES));
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trunk_split_c128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, XS_LDS_BYTES));
        c->split_attr_set = true;
    }
    const int grid = (B + XS_P - 1) / XS_P;
    if (halves_dtype == CZ_F16)
        hipLaunchKernelGGL((k_trunk_split_c128<true>), dim3(grid), dim3(XS_THREADS), XS_LDS_BYTES, c->stream, (const uint16_t *)wpk, bias,
                           trunk_out, head_w, head_b, head_out, (const uint16_t *)planes16, (const uint16_t *)w0, b0, B, 2 * nblocks,
                           c->batch_count, clock_probe(c, grid));
    else
        hipLaunchKernelGGL((k_trunk_split_c128<false>), dim3(grid), dim3(XS_THREADS), XS_LDS_BYTES, c->stream, (const uint16_t *)wpk, bias,
                           trunk_out, head_w, head_b, head_out, (const uint16_t *)planes16, (const uint16_t *)w0, b0, B, 2 * nblocks,
                           c->batch_count, clock_probe(c, grid));
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

extern "C" int cz_net_trunk_mx(cz_ctx *c, const void *planes16, const void *w0, const float *b0, const void *wpk, const float *bias,
                               float *trunk_out, const float *head_w, const float *head_b, float *head_out, int B, int nblocks) {
    using namespace czconv;
    CZ_REQUIRE(c && planes16 && w0 && b0 && wpk && bias && B >= 0 && nblocks >= 1 && (trunk_out || head_out),
               "cz_net_trunk_mx: null argument / nblocks < 1");
    CZ_REQUIRE(!head_out || (head_w && head_b), "cz_net_trunk_mx: head_out needs head_w and head_b");
    if (head_w && (reinterpret_cast<uintptr_t>(head_w) & 15u)) { cz_set_error("cz_net_trunk_mx: head_w must be 16-byte aligned"); return CZ_EINVAL; }
    if (reinterpret_cast<uintptr_t>(wpk) & 15u) { cz_set_error("cz_net_trunk_mx: wpk must be 16-byte aligned"); return CZ_EINVAL; }
    if (B == 0) return CZ_OK;
    const int grid = (B + MX_P - 1) / MX_P;
#ifdef CZ_EXPERIMENT_MX12
    {
        const char *e = getenv("CCHESS_MX_KERNEL");
        if (e && e[0] == '1' && e[1] == '2') {
            if (!c->mx2_attr_set) {
                CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trunk_mx12_c128), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES));
                c->mx2_attr_set = true;
            }
            hipLaunchKernelGGL(k_trunk_mx12_c128, dim3(grid), dim3(768), MX_LDS_BYTES, c->stream, (const unsigned char *)wpk, bias, trunk_out,
                               head_w, head_b, head_out, (const uint16_t *)planes16, (const uint16_t *)w0, b0, B, 2 * nblocks, c->batch_count,
                               clock_probe(c, grid));
            CZ_HIP(hipGetLastError());
            return CZ_OK;
        }
    }
#endif
#ifdef CZ_EXPERIMENT_MX2
    if (!c->mx_kernel) {   // experiment builds only: CCHESS_MX_KERNEL=2 selects k_trunk_mx2_c128
        const char *e = getenv("CCHESS_MX_KERNEL");
        c->mx_kernel = (e && e[0] == '2') ? 2 : 1;
    }
    if (c->mx_kernel == 2) {
        if (!c->mx2_attr_set) {
            CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trunk_mx2_c128), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES));
            c->mx2_attr_set = true;
        }
        const size_t need = (size_t)grid * MX2_XBUF_FLOATS_PER_WG * sizeof(float);
        if (need > c->mx_xbuf_bytes) {   // grown outside any capture: the first launch of a batch size allocates (as torch's allocator would)
            CZ_HIP(hipStreamSynchronize(c->stream));
            if (c->mx_xbuf) (void)hipFree(c->mx_xbuf);
            c->mx_xbuf = nullptr; c->mx_xbuf_bytes = 0;
            if (hipMalloc(&c->mx_xbuf, need) != hipSuccess) { c->mx_xbuf = nullptr; cz_set_error("cz_net_trunk_mx: hipMalloc(%zu B) for the block-input scratch failed", need); return CZ_ENOMEM; }
            c->mx_xbuf_bytes = need;
        }
        hipLaunchKernelGGL(k_trunk_mx2_c128, dim3(grid), dim3(MX_THREADS), MX_LDS_BYTES, c->stream, (const unsigned char *)wpk, bias, trunk_out,
                           head_w, head_b, head_out, (const uint16_t *)planes16, (const uint16_t *)w0, b0, B, 2 * nblocks, c->batch_count,
                           clock_probe(c, grid), (float *)c->mx_xbuf);
        CZ_HIP(hipGetLastError());
        return CZ_OK;
    }
#endif
    if (!c->mx_attr_set) {
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_trunk_mx_c128), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES));
        c->mx_attr_set = true;
    }
    hipLaunchKernelGGL(k_trunk_mx_c128, dim3(grid), dim3(MX_THREADS), MX_LDS_BYTES, c->stream, (const unsigned char *)wpk, bias, trunk_out,
                       head_w, head_b, head_out, (const uint16_t *)planes16, (const uint16_t *)w0, b0, B, 2 * nblocks, c->batch_count,
                       clock_probe(c, grid));
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

extern "C" int cz_set_clock_probe(cz_ctx *c, unsigned long long *buf_dev, int max_workgroups) {
    CZ_REQUIRE(c && max_workgroups >= 0, "cz_set_clock_probe: null context");
    c->clock_probe = buf_dev;
    c->clock_probe_wgs = buf_dev ? max_workgroups : 0;
    c->clock_probe_last_grid = 0;
    return CZ_OK;
}

extern "C" int cz_clock_probe_last_grid(cz_ctx *c) { return c ? c->clock_probe_last_grid : 0; }
