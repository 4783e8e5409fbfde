// How fast can a CU pull GEMM operand tiles from L2, and does it matter where they land?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/fill_rate scripts/fill_rate.hip && scripts/fill_rate
// 256 workgroups of 512 threads (one per CU), the M = 512, K = N = 8192 INT8 GEMM's access pattern: a
// workgroup walks K in 128-byte steps over a 128-row panel (row stride 8192 bytes); "A-like" panels are
// shared by 64 workgroups (4 panels, 4 MB: cache-resident), "B-like" panels by 4 workgroups of one XCD
// (64 panels, 64 MB: streamed from HBM once).  Modes: tiles to LDS by LDS-DMA (ring of 4 stages, counted
// waits), tiles to registers by plain 16-byte loads (3 steps in flight), and both at once (A to LDS, B to
// registers), which is what a kernel would do that keeps the weight operand out of LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int K = 8192, STEPS = K / 128;

// mode bit 0: A panel by LDS-DMA, bit 1: B panel by LDS-DMA, bit 2: B panel to registers (fragment-shaped:
// 16 rows x 64 B per instruction), bit 3: B panel to registers in 8-row x 128-byte pieces
template <int MODE>
__global__ __launch_bounds__(512) void fill(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, int* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int orig = blockIdx.x, xcd = orig & 7, wg = xcd * 32 + (orig >> 3);   // 256 workgroups, 32 per XCD
    const int bm = wg & 3, bn = wg >> 2;
    const uint8_t* a_base = A + (size_t)bm * 128 * K;
    const uint8_t* b_base = B + (size_t)bn * 128 * K;
    unsigned src[2];
    int dst[2];
    for (int j = 0; j < 2; j++) {
        const int pi = w + 8 * j, row = pi * 8 + (lane >> 3), slot = lane & 7;
        src[j] = (unsigned)row * K + slot * 16;
        dst[j] = pi * 1024;
    }
    // fragment-shaped source: wave w owns rows 16w..16w+15; lane (r, q) reads 16 bytes at k = 16q (+64 for the second slab)
    const unsigned fsrc = (unsigned)(w * 16 + (lane & 15)) * K + (lane >> 4) * 16;
    v4i acc = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; rep++) {
        auto issue_lds = [&](int kt) {
            uint8_t* st = lds + (kt & 3) * 32768;
            if (MODE & 1)
                for (int j = 0; j < 2; j++)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a_base + (size_t)kt * 128 + src[j]), (lptr_t)(st + dst[j]), 16, 0, 0);
            if (MODE & 2)
                for (int j = 0; j < 2; j++)
                    __builtin_amdgcn_global_load_lds((gptr_t)(b_base + (size_t)kt * 128 + src[j]), (lptr_t)(st + 16384 + dst[j]), 16, 0, 0);
        };
        auto load_reg = [&](int kt, v4i (&r)[2]) {
            if (MODE & 4) {
                r[0] = *reinterpret_cast<const v4i*>(b_base + (size_t)kt * 128 + fsrc);
                r[1] = *reinterpret_cast<const v4i*>(b_base + (size_t)kt * 128 + fsrc + 64);
            } else if (MODE & 8) {
                r[0] = *reinterpret_cast<const v4i*>(b_base + (size_t)kt * 128 + src[0]);
                r[1] = *reinterpret_cast<const v4i*>(b_base + (size_t)kt * 128 + src[1]);
            }
        };
        if constexpr ((MODE & 12) == 0) {
            constexpr int per = ((MODE & 1) ? 2 : 0) + ((MODE & 2) ? 2 : 0);
            issue_lds(0); issue_lds(1); issue_lds(2);
            for (int kt = 0; kt < STEPS; kt++) {
                // tile kt+1 landed, two later tiles stay in flight; one barrier per step as the GEMM has
                if (per == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_lds(min(kt + 3, STEPS - 1));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            acc[0] ^= *reinterpret_cast<const int*>(lds + t * 4);
        } else if constexpr ((MODE & 3) == 0) {
            v4i r0[2], r1[2], r2[2], r3[2];
            load_reg(0, r0); load_reg(1, r1); load_reg(2, r2);
            for (int kt = 0; kt < STEPS; kt += 4) {
                load_reg(min(kt + 3, STEPS - 1), r3); acc ^= r0[0] ^ r0[1]; __builtin_amdgcn_s_barrier();
                load_reg(min(kt + 4, STEPS - 1), r0); acc ^= r1[0] ^ r1[1]; __builtin_amdgcn_s_barrier();
                load_reg(min(kt + 5, STEPS - 1), r1); acc ^= r2[0] ^ r2[1]; __builtin_amdgcn_s_barrier();
                load_reg(min(kt + 6, STEPS - 1), r2); acc ^= r3[0] ^ r3[1]; __builtin_amdgcn_s_barrier();
            }
            acc ^= r0[0] ^ r1[0] ^ r2[0];
        } else {
            // A by LDS-DMA (2 requests per step and wave), B to registers by inline-asm loads so that both
            // queues are counted by hand: per step 2 DMA requests then 2 register loads
            v4i r[4][2];
            auto load_asm = [&](int kt, v4i (&x)[2]) {
                const uint8_t* p = b_base + (size_t)kt * 128 + ((MODE & 4) ? fsrc : src[0]);
                const uint8_t* p2 = (MODE & 4) ? p + 64 : b_base + (size_t)kt * 128 + src[1];
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[0]) : "v"(p) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[1]) : "v"(p2) : "memory");
            };
            issue_lds(0); load_asm(0, r[0]);
            issue_lds(1); load_asm(1, r[1]);
            issue_lds(2); load_asm(2, r[2]);
            for (int kt = 0; kt < STEPS; kt += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    // outstanding: 3 steps x 4; step kt+u complete when <= 8 remain
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    acc ^= r[u][0] ^ r[u][1];
                    issue_lds(min(kt + u + 3, STEPS - 1));
                    load_asm(min(kt + u + 3, STEPS - 1), r[(u + 3) & 3]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc ^= r[0][0] ^ r[1][0] ^ r[2][0];
            __builtin_amdgcn_s_barrier();
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[0] = 1;
}

template <int MODE>
void run(const char* name, const uint8_t* A, const uint8_t* B, int* out, double bytes_per_cu_step) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(fill<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 1, launches = 20;
    fill<MODE><<<256, 512, 131072>>>(A, B, out, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; i++) fill<MODE><<<256, 512, 131072>>>(A, B, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / launches;
    const double per_cu = bytes_per_cu_step * STEPS * reps;
    printf("%-58s %7.1f us per launch  %6.1f GB/s per CU  %5.1f TB/s chip\n", name, us, per_cu / (us * 1e-6) / 1e9,
           per_cu * 256 / (us * 1e-6) / 1e12);
}

int main() {
    uint8_t *A, *B;
    int* out;
    hipMalloc(&A, (size_t)512 * K);
    hipMalloc(&B, (size_t)8192 * K);
    hipMalloc(&out, 16);
    hipMemset(A, 1, (size_t)512 * K);
    hipMemset(B, 2, (size_t)8192 * K);
    run<1>("A panel -> LDS (DMA)", A, B, out, 16384);
    run<2>("B panel -> LDS (DMA)", A, B, out, 16384);
    run<3>("A + B -> LDS (DMA), the GEMM's fill", A, B, out, 32768);
    run<4>("B -> registers, 16 rows x 64 B per instruction", A, B, out, 16384);
    run<8>("B -> registers, 8 rows x 128 B per instruction", A, B, out, 16384);
    run<5>("A -> LDS (DMA) + B -> registers (16 x 64 B)", A, B, out, 32768);
    run<9>("A -> LDS (DMA) + B -> registers (8 x 128 B)", A, B, out, 32768);
    return 0;
}
