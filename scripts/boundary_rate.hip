// What a dependent kernel boundary costs on this GPU inside a captured HIP graph: a chain of tiny kernels, each reading
// what the previous one wrote; every workgroup stamps its entry and exit with the 100 MHz wall clock that all CUs
// share.  boundary = first entry of launch k+1 - last exit of launch k; also the host-visible period per launch.
//   boundary_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Big { int pad[240]; };   // 960 bytes of kernel arguments, like a batched-launch table

template <int LDSKB, bool BIGARG>
__global__ __launch_bounds__(256) void hop(const int* __restrict__ in, int* __restrict__ out, unsigned long long* tr, int launch,
                                           int nwg, Big big) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ int lds[LDSKB ? LDSKB * 256 : 1];
    int v = in[threadIdx.x & 63];
    if (BIGARG) v += big.pad[threadIdx.x % 240];
    if (LDSKB) { lds[threadIdx.x] = v; __syncthreads(); v = lds[(threadIdx.x + 1) & 255]; }
    out[blockIdx.x * 64 + (threadIdx.x & 63)] = v + 1;
    if (threadIdx.x == 0) {
        tr[((size_t)launch * nwg + blockIdx.x) * 2] = t0;
        tr[((size_t)launch * nwg + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int LDSKB, bool BIGARG>
void run(const char* name, int nwg) {
    const int L = 64;
    int *a, *b;
    unsigned long long* tr;
    CK(hipMalloc(&a, nwg * 64 * 4 + 256)); CK(hipMalloc(&b, nwg * 64 * 4 + 256));
    CK(hipMemset(a, 0, nwg * 64 * 4 + 256)); CK(hipMemset(b, 0, nwg * 64 * 4 + 256));
    CK(hipMalloc(&tr, (size_t)L * nwg * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    Big big = {};
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < L; i++) hop<LDSKB, BIGARG><<<nwg, 256, 0, st>>>(i & 1 ? b : a, i & 1 ? a : b, tr, i, nwg, big);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)L * nwg * 2);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> gaps, spans;
    for (int i = 1; i < L; i++) {
        unsigned long long first = ~0ull, last = 0, lastprev = 0;
        for (int w = 0; w < nwg; w++) {
            first = std::min(first, h[((size_t)i * nwg + w) * 2]);
            last = std::max(last, h[((size_t)i * nwg + w) * 2 + 1]);
            lastprev = std::max(lastprev, h[((size_t)(i - 1) * nwg + w) * 2 + 1]);
        }
        gaps.push_back(((double)first - (double)lastprev) / 100.0);
        spans.push_back(((double)last - (double)first) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end()); std::sort(spans.begin(), spans.end());
    printf("%-34s wgs %5d: boundary (last exit -> next first entry) median %.2f us (min %.2f), in-kernel span %.2f us, period %.2f us/launch\n",
           name, nwg, gaps[gaps.size() / 2], gaps[0], spans[spans.size() / 2], ms * 1e3 / (10.0 * L));
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(tr));
}

int main() {
    for (int nwg : {64, 256, 1024}) {
        run<0, false>("plain", nwg);
        run<32, false>("32 KB LDS", nwg);
        run<0, true>("960 B of arguments", nwg);
    }
    return 0;
}
