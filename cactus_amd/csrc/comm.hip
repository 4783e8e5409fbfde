// Multi-GPU exchange step of the N-split matmuls and the kv-head-sharded attention: one process per GPU, RCCL over
// xGMI.  The reference has no multi-device code at all (a CPU engine); this is the partition of its matmul by
// weight rows: rank r owns rows [start, end) of B[N, K] and of its scales - a contiguous slice of the reference's
// own layout - computes C[:, start:end], and ONE exchange makes every rank hold C[M, N].  No output element's sum
// is split, so the gathered result equals the single-GPU result bit for bit.
//
// Exchange forms (everything is enqueued on the caller's stream, allocates nothing per call, and can be captured
// into the step's HIP graph):
//   M == 1 (decode): the slices are contiguous in C, so rank-major IS column order: an in-place ncclAllGather when
//     every rank owns the same count, otherwise (or on request) grouped ncclSend / ncclRecv of each slice to / from
//     every peer - the direct all-pairs schedule that uses all 7 xGMI links of a GPU at once for KB-sized messages.
//   M > 1: a rank's block [M, count] is strided in C[M, N]; RCCL moves contiguous runs only and one message per row
//     segment (M x (world - 1) messages of a few KB) would be bound by per-message cost.  The local GEMM therefore
//     writes its block contiguously into its slot of a rank-major staging area [world][M][per] in the library
//     workspace (no padding copy), one in-place ncclAllGather fills the other slots, and ONE kernel copies the
//     slots to their columns of C (a pass over M x N halves, a few microseconds against the gather itself).
//
// RCCL is loaded at first use (dlopen of librccl.so.1, which finds the copy a host process such as PyTorch has
// already loaded), so the library itself loads on machines without it.
#include "../../include/cactus_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "kernels.h"

namespace cactus_hip {
void set_last_error(const char* msg);
int workspace_for_comm(size_t bytes, void** out);   // api.hip: the grow-only device workspace
}  // namespace cactus_hip

using namespace cactus_hip;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        g_rccl_error = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "dlopen failed");
        return;
    }
#define SYM(field, name)                                                                   \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name));   \
    if (!g_rccl.field) g_rccl_error = std::string("RCCL lacks ") + name;
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
}

int fail(int code, const std::string& what) {
    set_last_error(what.c_str());
    return code;
}

bool rccl_ready() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle != nullptr && g_rccl_error.empty();
}

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(CACTUS_HIP_ERUNTIME, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

struct Comm {
    ncclComm_t nccl = nullptr;
    int world = 1, rank = 0;
    bool owned = true;
};

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// slots [world][M][per] -> C[M][N]: thread per 16 bytes of a row segment (per and every column start are multiples of
// 8 halves when `vec`), else per half
__global__ __launch_bounds__(256) void slots_to_columns_kernel(const uint16_t* __restrict__ slots, uint16_t* __restrict__ C,
                                                               size_t M, size_t N, size_t per, int world, int vec) {
    const size_t step = vec ? 8 : 1;
    const size_t per_v = per / step;
    const size_t total = (size_t)world * M * per_v;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t c = (i % per_v) * step, m = (i / per_v) % M, r = i / (per_v * M);
        const size_t col = r * per + c;
        if (col >= N) continue;
        const uint16_t* src = slots + (r * M + m) * per + c;
        uint16_t* dst = C + m * N + col;
        if (vec && col + 8 <= N) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
        } else {
            for (size_t e = 0; e < step && col + e < N; e++) dst[e] = src[e];
        }
    }
}

size_t shard_per(size_t N, int world, size_t multiple) {
    size_t per = (N + (size_t)world - 1) / (size_t)world;
    if (multiple > 1) per = (per + multiple - 1) / multiple * multiple;
    return per;
}

hipError_t slots_to_columns(const uint16_t* slots, uint16_t* C, size_t M, size_t N, size_t per, int world, hipStream_t st) {
    const int vec = (per % 8 == 0) && (N % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(slots) & 15) == 0);
    const size_t total = (size_t)world * M * (per / (vec ? 8 : 1));
    if (total == 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 16384);
    slots_to_columns_kernel<<<grid, 256, 0, st>>>(slots, C, M, N, per, world, vec);
    return hipGetLastError();
}

// M == 1 (or any case where every rank's columns are one contiguous run of C): exchange in place
int gather_row(Comm* c, uint16_t* C, size_t N, size_t per, int algo, hipStream_t st) {
    const bool uniform = per * (size_t)c->world == N;
    if (c->world == 1) return CACTUS_HIP_OK;
    if (uniform && algo != 2) {
        RCCL_TRY(g_rccl.AllGather(C + (size_t)c->rank * per, C, per, ncclHalf, c->nccl, st));
        return CACTUS_HIP_OK;
    }
    // direct all-pairs: my slice to every peer, every peer's slice from it
    const size_t my0 = std::min(N, (size_t)c->rank * per), my1 = std::min(N, my0 + per);
    RCCL_TRY(g_rccl.GroupStart());
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank) continue;
        const size_t p0 = std::min(N, (size_t)p * per), p1 = std::min(N, p0 + per);
        if (my1 > my0) RCCL_TRY(g_rccl.Send(C + my0, my1 - my0, ncclHalf, p, c->nccl, st));
        if (p1 > p0) RCCL_TRY(g_rccl.Recv(C + p0, p1 - p0, ncclHalf, p, c->nccl, st));
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return CACTUS_HIP_OK;
}

int matmul_nsplit(bool int4, Comm* c, const int8_t* A, const float* As, const void* B_rows, const cactus_half_t* Bs_rows,
                  cactus_half_t* C, size_t M, size_t K, size_t N, size_t gs, size_t multiple, int algo, void* stream) {
    if (!c) return fail(CACTUS_HIP_EINVAL, "matmul_nsplit: null communicator");
    if (M == 0 || K == 0 || N == 0) return CACTUS_HIP_OK;
    if (!C) return fail(CACTUS_HIP_EINVAL, "matmul_nsplit: null C");
    if (multiple == 0) multiple = 16;
    const size_t per = shard_per(N, c->world, multiple);
    const size_t n0 = std::min(N, (size_t)c->rank * per), n1 = std::min(N, n0 + per);
    const size_t rows = n1 - n0;
    auto local = [&](cactus_half_t* dst) -> int {
        if (rows == 0) return CACTUS_HIP_OK;
        if (!B_rows || !Bs_rows) return fail(CACTUS_HIP_EINVAL, "matmul_nsplit: null weight slice");
        return int4 ? cactus_hip_matmul_int4_dev(A, As, static_cast<const uint8_t*>(B_rows), Bs_rows, dst, M, K, rows, gs, stream)
                    : cactus_hip_matmul_int8_dev(A, As, static_cast<const int8_t*>(B_rows), Bs_rows, dst, M, K, rows, gs, stream);
    };
    if (M == 1) {
        int rc = local(C + n0);
        if (rc != CACTUS_HIP_OK) return rc;
        return gather_row(c, C, N, per, algo, S(stream));
    }
    if (c->world == 1) return local(C);
    void* ws = nullptr;
    int rc = workspace_for_comm((size_t)c->world * M * per * sizeof(uint16_t), &ws);
    if (rc != CACTUS_HIP_OK) return rc;
    uint16_t* slots = static_cast<uint16_t*>(ws);
    uint16_t* mine = slots + (size_t)c->rank * M * per;
    if (rows == per) {
        rc = local(mine);                       // the block is the slot: [M, per] contiguous
        if (rc != CACTUS_HIP_OK) return rc;
    } else if (rows > 0) {
        // a short last shard: its [M, rows] block goes to the front of the slot row by row (rows < per)
        rc = local(C);                          // C[0 .. M * rows) as scratch: overwritten by the copy-out below
        if (rc != CACTUS_HIP_OK) return rc;
        hipError_t e = hipMemcpy2DAsync(mine, per * 2, C, rows * 2, rows * 2, M, hipMemcpyDeviceToDevice, S(stream));
        if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, std::string("hipMemcpy2DAsync: ") + hipGetErrorString(e));
    }
    RCCL_TRY(g_rccl.AllGather(mine, slots, M * per, ncclHalf, c->nccl, S(stream)));
    hipError_t e = slots_to_columns(slots, C, M, N, per, c->world, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, std::string("slots_to_columns launch: ") + hipGetErrorString(e));
    return CACTUS_HIP_OK;
}

}  // namespace

extern "C" {

int cactus_hip_comm_available(void) { return rccl_ready() ? 1 : 0; }

int cactus_hip_comm_get_unique_id(void* id128) {
    if (!id128) return fail(CACTUS_HIP_EINVAL, "comm_get_unique_id: null buffer");
    if (!rccl_ready()) return fail(CACTUS_HIP_ERUNTIME, g_rccl_error);
    static_assert(sizeof(ncclUniqueId) == CACTUS_HIP_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return CACTUS_HIP_OK;
}

int cactus_hip_comm_init(void** comm_out, const void* id128, int world, int rank) {
    if (!comm_out || !id128 || world < 1 || rank < 0 || rank >= world)
        return fail(CACTUS_HIP_EINVAL, "comm_init: bad arguments");
    if (!rccl_ready()) return fail(CACTUS_HIP_ERUNTIME, g_rccl_error);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    Comm* c = new Comm;
    c->world = world;
    c->rank = rank;
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(CACTUS_HIP_ERUNTIME, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    }
    *comm_out = c;
    return CACTUS_HIP_OK;
}

int cactus_hip_comm_adopt(void** comm_out, void* nccl_comm, int world, int rank) {
    if (!comm_out || !nccl_comm || world < 1 || rank < 0 || rank >= world)
        return fail(CACTUS_HIP_EINVAL, "comm_adopt: bad arguments");
    if (!rccl_ready()) return fail(CACTUS_HIP_ERUNTIME, g_rccl_error);
    Comm* c = new Comm;
    c->nccl = static_cast<ncclComm_t>(nccl_comm);
    c->world = world;
    c->rank = rank;
    c->owned = false;
    *comm_out = c;
    return CACTUS_HIP_OK;
}

int cactus_hip_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return CACTUS_HIP_OK;
    if (c->owned && c->nccl && rccl_ready()) (void)g_rccl.CommDestroy(c->nccl);
    delete c;
    return CACTUS_HIP_OK;
}

int cactus_hip_comm_rank(void* comm) { return comm ? static_cast<Comm*>(comm)->rank : -1; }
int cactus_hip_comm_world(void* comm) { return comm ? static_cast<Comm*>(comm)->world : -1; }

void cactus_hip_row_shard(size_t N, int world, int rank, size_t multiple, size_t* start, size_t* end) {
    const size_t per = shard_per(N, world < 1 ? 1 : world, multiple ? multiple : 1);
    const size_t s = std::min(N, (size_t)(rank < 0 ? 0 : rank) * per);
    if (start) *start = s;
    if (end) *end = std::min(N, s + per);
}

int cactus_hip_slots_to_columns_dev(const cactus_half_t* slots, cactus_half_t* C, size_t M, size_t N, int world,
                                    size_t multiple, void* stream) {
    if (M == 0 || N == 0) return CACTUS_HIP_OK;
    if (!slots || !C || world < 1) return fail(CACTUS_HIP_EINVAL, "slots_to_columns: bad arguments");
    const size_t per = shard_per(N, world, multiple ? multiple : 1);
    hipError_t e = slots_to_columns(slots, C, M, N, per, world, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, std::string("slots_to_columns launch: ") + hipGetErrorString(e));
    return CACTUS_HIP_OK;
}

int cactus_hip_allgather_columns_dev(void* comm, cactus_half_t* C, size_t M, size_t N, size_t multiple, int algo,
                                     void* stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return fail(CACTUS_HIP_EINVAL, "allgather_columns: null communicator");
    if (M == 0 || N == 0) return CACTUS_HIP_OK;
    if (!C) return fail(CACTUS_HIP_EINVAL, "allgather_columns: null C");
    if (c->world > 1 && !rccl_ready()) return fail(CACTUS_HIP_ERUNTIME, g_rccl_error);
    const size_t per = shard_per(N, c->world, multiple ? multiple : 1);
    if (M == 1) return gather_row(c, C, N, per, algo, S(stream));
    if (c->world == 1) return CACTUS_HIP_OK;
    // M > 1: this rank's columns of C -> its slot, gather, slots -> columns
    void* ws = nullptr;
    int rc = workspace_for_comm((size_t)c->world * M * per * sizeof(uint16_t), &ws);
    if (rc != CACTUS_HIP_OK) return rc;
    uint16_t* slots = static_cast<uint16_t*>(ws);
    uint16_t* mine = slots + (size_t)c->rank * M * per;
    const size_t n0 = std::min(N, (size_t)c->rank * per), n1 = std::min(N, n0 + per);
    if (n1 > n0) {
        hipError_t e = hipMemcpy2DAsync(mine, per * 2, C + n0, N * 2, (n1 - n0) * 2, M, hipMemcpyDeviceToDevice, S(stream));
        if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, std::string("hipMemcpy2DAsync: ") + hipGetErrorString(e));
    }
    RCCL_TRY(g_rccl.AllGather(mine, slots, M * per, ncclHalf, c->nccl, S(stream)));
    hipError_t e = slots_to_columns(slots, C, M, N, per, c->world, S(stream));
    if (e != hipSuccess) return fail(CACTUS_HIP_ERUNTIME, std::string("slots_to_columns launch: ") + hipGetErrorString(e));
    return CACTUS_HIP_OK;
}

int cactus_hip_matmul_int8_nsplit_dev(void* comm, const int8_t* A, const float* A_scales, const int8_t* B_rows,
                                      const cactus_half_t* B_scales_rows, cactus_half_t* C, size_t M, size_t K, size_t N,
                                      size_t group_size, size_t multiple, int algo, void* stream) {
    return matmul_nsplit(false, static_cast<Comm*>(comm), A, A_scales, B_rows, B_scales_rows, C, M, K, N, group_size,
                         multiple, algo, stream);
}
int cactus_hip_matmul_int4_nsplit_dev(void* comm, const int8_t* A, const float* A_scales, const uint8_t* B_rows_packed,
                                      const cactus_half_t* B_scales_rows, cactus_half_t* C, size_t M, size_t K, size_t N,
                                      size_t group_size, size_t multiple, int algo, void* stream) {
    return matmul_nsplit(true, static_cast<Comm*>(comm), A, A_scales, B_rows_packed, B_scales_rows, C, M, K, N, group_size,
                         multiple, algo, stream);
}

}  // extern "C"
