// Yardsticks for a launch of the step's size and traffic shape on this box (development aid, no product code):
//   empty      back-to-back launches of an empty kernel of G x B threads: launch boundary + dispatch ramp
//   rows       W waves, each: read one 2 KB table row at a pseudo-random address + C gradient rows (streamed), write the
//              row back and M output rows (streamed) -- the step's real traffic (42.7 MB at W = 6000, C = 1.1, M = 1.1)
//              WITHOUT any metadata chain (addresses from the wave number).  ipw = items per wave with all loads of
//              both items in flight at once.
// usage: floor_bench [table_GiB=16]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

struct RowArgs {
    float *table; uint64_t rows;
    const float *grads; float *out;
    int items;      // total items
    int c, m;       // gradient rows read / output rows written per item
    uint32_t seed;
};
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int IPW, int C, int M, int FLAGS = 0>
__global__ __launch_bounds__(1024, 8 / IPW) void rows_kernel(const RowArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    f4 r0[IPW], r1[IPW], g0[IPW][C], g1[IPW][C];
    float *row[IPW], *wrow[IPW];
    int item[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        item[k] = __builtin_amdgcn_readfirstlane(wave * IPW + k);
        const uint64_t r = (uint64_t)mix(item[k] * 2654435761u + a.seed) % a.rows;
        row[k] = a.table + r * 512;
        wrow[k] = (FLAGS & 1) ? a.table + ((uint64_t)mix(item[k] * 40503u + a.seed + 17u) % a.rows) * 512 : row[k];
    }
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        if (item[k] < a.items) {
            if (FLAGS & 4) {
                r0[k] = __builtin_nontemporal_load((const f4 *)(row[k] + 4 * lane));
                r1[k] = __builtin_nontemporal_load((const f4 *)(row[k] + 256 + 4 * lane));
            } else {
                r0[k] = *(const f4 *)(row[k] + 4 * lane);
                r1[k] = *(const f4 *)(row[k] + 256 + 4 * lane);
            }
#pragma unroll
            for (int t = 0; t < C; ++t) {
                const float *g = a.grads + ((uint64_t)item[k] * C + t) * 512;
                g0[k][t] = *(const f4 *)(g + 4 * lane);
                g1[k][t] = *(const f4 *)(g + 256 + 4 * lane);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        if (item[k] < a.items) {
#pragma unroll
            for (int t = 0; t < C; ++t) { r0[k] -= 0.001f * g0[k][t]; r1[k] -= 0.001f * g1[k][t]; }
            if (FLAGS & 2) {
                *(f4 *)(wrow[k] + 4 * lane) = r0[k];
                *(f4 *)(wrow[k] + 256 + 4 * lane) = r1[k];
            } else {
                __builtin_nontemporal_store(r0[k], (f4 *)(wrow[k] + 4 * lane));
                __builtin_nontemporal_store(r1[k], (f4 *)(wrow[k] + 256 + 4 * lane));
            }
#pragma unroll
            for (int t = 0; t < M; ++t) {
                float *o = a.out + ((uint64_t)item[k] * M + t) * 512;
                if (FLAGS & 8) {
                    *(f4 *)(o + 4 * lane) = r0[k];
                    *(f4 *)(o + 256 + 4 * lane) = r1[k];
                } else {
                    __builtin_nontemporal_store(r0[k], (f4 *)(o + 4 * lane));
                    __builtin_nontemporal_store(r1[k], (f4 *)(o + 256 + 4 * lane));
                }
            }
        }
    }
}

static float time_launches(hipStream_t s, int reps, const std::function<void(int)> &launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}
#include <functional>

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const uint64_t rows = (uint64_t)(gib * (1ull << 30) / 2048);
    hipStream_t s; CK(hipStreamCreate(&s));
    float *table; CK(hipMalloc(&table, rows * 2048));
    CK(hipMemsetAsync(table, 0, rows * 2048, s));
    const int NB = 24, items_max = 8192;
    std::vector<float *> grads(NB), outs(NB);
    for (int i = 0; i < NB; ++i) {
        CK(hipMalloc(&grads[i], (size_t)items_max * 2 * 2048)); CK(hipMemsetAsync(grads[i], 0, (size_t)items_max * 2 * 2048, s));
        CK(hipMalloc(&outs[i], (size_t)items_max * 2 * 2048));
    }
    CK(hipStreamSynchronize(s));
    printf("# empty kernel, back-to-back on one stream: us per launch\n");
    const int shapes[][2] = {{1, 64}, {512, 1024}, {448, 1024}, {384, 1024}, {256, 1024}, {128, 1024}, {2048, 256}, {1024, 256}, {1024, 512}, {8192, 64}, {4096, 64}};
    for (auto &sh : shapes) {
        const float us = time_launches(s, 2000, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, s, (int *)nullptr); });
        printf("empty grid %5d x %4d threads (%5d waves): %6.2f us\n", sh[0], sh[1], sh[0] * sh[1] / 64, us);
    }
    printf("# rows kernel: items x (1 row rw + C grad rows read + M out rows written), 2 KB rows, table %.0f GiB\n", gib);
    auto run = [&](const char *name, int items, int ipw, int c, int m, int wg_threads) {
        RowArgs a{table, rows, nullptr, nullptr, items, c, m, 0};
        const int waves = (items + ipw - 1) / ipw, wpw = wg_threads / 64;
        const int grid = (waves + wpw - 1) / wpw;
        const float us = time_launches(s, 1000, [&](int i) {
            RowArgs b = a; b.grads = grads[i % NB]; b.out = outs[i % NB]; b.seed = i * 7919u;
#define L(I, C_, M_) hipLaunchKernelGGL((rows_kernel<I, C_, M_>), dim3(grid), dim3(wg_threads), 0, s, b)
            if (ipw == 1 && c == 1 && m == 1) L(1, 1, 1);
            else if (ipw == 2 && c == 1 && m == 1) L(2, 1, 1);
            else if (ipw == 1 && c == 2 && m == 2) L(1, 2, 2);
            else if (ipw == 2 && c == 2 && m == 2) L(2, 2, 2);
            else if (ipw == 1 && c == 0 && m == 1) L(1, 0, 1);
            else if (ipw == 1 && c == 1 && m == 0) L(1, 1, 0);
#undef L
        });
        const double bytes = (double)items * 2048.0 * (2 + c + m);
        printf("%-34s items %5d ipw %d C %d M %d wg %4d: %6.2f us/launch  %.2f MB  %.2f TB/s\n", name, items, ipw, c, m,
               wg_threads, us, bytes / 1e6, bytes / us / 1e6);
    };
    for (int wg : {1024, 256}) {
        run("step-like (C=1,M=1)", 5200, 1, 1, 1, wg);      // 5200 * 4 * 2 KB = 42.6 MB
        run("step-like two items per wave", 5200, 2, 1, 1, wg);
        run("heavier items (C=2,M=2)", 3470, 1, 2, 2, wg);  // same bytes, fewer waves
        run("heavier, two per wave", 3470, 2, 2, 2, wg);
        run("copy only (C=0,M=1)", 6940, 1, 0, 1, wg);
        run("apply only (C=1,M=0)", 6940, 1, 1, 0, wg);
    }
    // ---- P independent chains: the step's items cut into P key classes, class p's launches back to back on stream p (a key's
    // items of consecutive steps fall into the same class, so the chains never wait for each other): us per STEP = all P launches
    printf("# P independent chains of step-like launches (5200 / P items each, 1024-thread workgroups): us per step\n");
    for (int P : {1, 2, 3, 4}) {
        std::vector<hipStream_t> ss(P);
        for (int p = 0; p < P; ++p) CK(hipStreamCreateWithFlags(&ss[p], hipStreamNonBlocking));
        const int items = 5200 / P, grid = (items + 15) / 16, reps = 1000;
        auto launch_all = [&](int i) {
            for (int p = 0; p < P; ++p) {
                RowArgs b{table, rows, grads[i % NB] + (size_t)p * items * 512, outs[i % NB] + (size_t)p * items * 512, items, 1, 1,
                          (uint32_t)(i * 7919u + p * 104729u)};
                hipLaunchKernelGGL((rows_kernel<1, 1, 1>), dim3(grid), dim3(1024), 0, ss[p], b);
            }
        };
        for (int i = 0; i < 20; ++i) launch_all(i);
        for (int p = 0; p < P; ++p) CK(hipStreamSynchronize(ss[p]));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<hipEvent_t> ej(P);
        for (int p = 0; p < P; ++p) CK(hipEventCreateWithFlags(&ej[p], hipEventDisableTiming));
        CK(hipEventRecord(e0, ss[0]));
        for (int p = 1; p < P; ++p) CK(hipStreamWaitEvent(ss[p], e0, 0));
        for (int i = 0; i < reps; ++i) launch_all(i);
        for (int p = 1; p < P; ++p) { CK(hipEventRecord(ej[p], ss[p])); CK(hipStreamWaitEvent(ss[0], ej[p], 0)); }
        CK(hipEventRecord(e1, ss[0]));
        CK(hipStreamSynchronize(ss[0]));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("chains %d: %6.2f us per step (%d items per launch)  %.2f TB/s\n", P, ms * 1000.f / reps, items,
               (double)items * P * 2048.0 * 4 / (ms * 1000.0 / reps) / 1e6);
    }
    printf("# variants of the step-like / apply-only kernels (256-thread workgroups unless said otherwise)\n");
    auto runf = [&](const char *name, int items, int c, int m, int wg, int flags) {
        RowArgs a{table, rows, nullptr, nullptr, items, c, m, 0};
        const int wpw = wg / 64, grid = (items + wpw - 1) / wpw;
        const float us = time_launches(s, 1000, [&](int i) {
            RowArgs b = a; b.grads = grads[i % NB]; b.out = outs[i % NB]; b.seed = i * 7919u;
#define LF(C_, M_, F) hipLaunchKernelGGL((rows_kernel<1, C_, M_, F>), dim3(grid), dim3(wg), 0, s, b)
            if (c == 1 && m == 1) { switch (flags) { case 0: LF(1, 1, 0); break; case 1: LF(1, 1, 1); break; case 2: LF(1, 1, 2); break;
                                                     case 4: LF(1, 1, 4); break; case 6: LF(1, 1, 6); break; case 8: LF(1, 1, 8); break; } }
            else if (c == 1 && m == 0) { switch (flags) { case 0: LF(1, 0, 0); break; case 1: LF(1, 0, 1); break; case 2: LF(1, 0, 2); break;
                                                          case 4: LF(1, 0, 4); break; case 6: LF(1, 0, 6); break; } }
#undef LF
        });
        const double bytes = (double)items * 2048.0 * (2 + c + m);
        printf("%-44s items %5d C %d M %d wg %4d flags %d: %6.2f us/launch  %.2f MB  %.2f TB/s\n", name, items, c, m, wg, flags, us,
               bytes / 1e6, bytes / us / 1e6);
    };
    for (int wg : {64, 128, 256, 512, 1024})
        runf("step-like by workgroup size", 5200, 1, 1, wg, 0);
    runf("step-like, row written to ANOTHER row", 5200, 1, 1, 256, 1);
    runf("step-like, plain row stores", 5200, 1, 1, 256, 2);
    runf("step-like, nt row loads", 5200, 1, 1, 256, 4);
    runf("step-like, nt row loads + plain row stores", 5200, 1, 1, 256, 6);
    runf("step-like, plain out stores", 5200, 1, 1, 256, 8);
    runf("apply-only", 6940, 1, 0, 256, 0);
    runf("apply-only, row written to ANOTHER row", 6940, 1, 0, 256, 1);
    runf("apply-only, plain row stores", 6940, 1, 0, 256, 2);
    runf("apply-only, nt row loads", 6940, 1, 0, 256, 4);
    runf("apply-only, nt row loads + plain row stores", 6940, 1, 0, 256, 6);
    return 0;
}
