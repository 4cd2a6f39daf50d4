// comm.hip — the one exchange step of the path: projected clip tokens between the ranks of a node, RCCL over xGMI.
//
// The reference never shards inside a sample; its closest ancestor is the accelerate all-gather of
// ref:scripts/general/generate_narration_texts.py:124-127.  Here every rank encodes the clips it was dealt and the
// (num_query x Dt) bf16 tokens of a clip (164 KB at OPT-2.7B) travel to the rank whose language-model pass consumes them:
//   eilev_gather_clip_tokens   — every rank receives every clip (ncclAllGather; latency mode, replicated LM)
//   eilev_exchange_clip_tokens — every rank receives only the clips of ITS samples (grouped ncclSend / ncclRecv = all-to-all-v)
// Both are asynchronous launches on the caller's stream (a side stream in eilev_amd/comm.py, so the exchange of one encode
// chunk runs under the ViT of the next).
//
// RCCL is bound at RUN time (dlopen + dlsym) from the library the process already has mapped — PyTorch ships its own
// librccl.so and two copies in one process must not meet — so libeilev_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

template <typename F>
bool sym(void *h, const char *name, F &out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

inline int rc(ncclResult_t r) { return r == ncclSuccess ? 0 : EILEV_E_RCCL_BASE + (int)r; }

}  // namespace

extern "C" {

int eilev_comm_bind(const char *librccl_path) {
    if (g_rccl.handle) return 0;
    const char *names[] = {librccl_path, "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // the copy already mapped into the process (torch's), if any
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return EILEV_E_UNSUPPORTED;
    Rccl r;
    r.handle = h;
    const bool ok = sym(h, "ncclGetUniqueId", r.GetUniqueId) && sym(h, "ncclCommInitRank", r.CommInitRank) &&
                    sym(h, "ncclCommDestroy", r.CommDestroy) && sym(h, "ncclAllGather", r.AllGather) &&
                    sym(h, "ncclSend", r.Send) && sym(h, "ncclRecv", r.Recv) && sym(h, "ncclGroupStart", r.GroupStart) &&
                    sym(h, "ncclGroupEnd", r.GroupEnd);
    sym(h, "ncclGetErrorString", r.GetErrorString);
    if (!ok) return EILEV_E_UNSUPPORTED;
    g_rccl = r;
    return 0;
}

int eilev_comm_unique_id(void *id) {
    if (!id) return EILEV_E_BADARG;
    if (!g_rccl.handle) return EILEV_E_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == EILEV_COMM_ID_BYTES, "ncclUniqueId size");
    return rc(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId *>(id)));
}

int eilev_comm_init(void **comm, int world, int rank, const void *id) {
    if (!comm || !id || world < 1 || rank < 0 || rank >= world) return EILEV_E_BADARG;
    if (!g_rccl.handle) return EILEV_E_UNSUPPORTED;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    const int e = rc(g_rccl.CommInitRank(&c, world, uid, rank));  // binds the CURRENT hip device
    if (e) return e;
    *comm = c;
    return 0;
}

int eilev_comm_destroy(void *comm) {
    if (!comm) return 0;
    if (!g_rccl.handle) return EILEV_E_UNSUPPORTED;
    return rc(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
}

// rows[r] rows of row_bytes from rank r land rank-major in `all`; `local` holds this rank's rows[rank] rows.
int eilev_gather_clip_tokens(void *comm, const void *local, void *all, const int64_t *rows, int world, int rank,
                             int64_t row_bytes, void *stream) {
    if (!all || !rows || world < 1 || rank < 0 || rank >= world || row_bytes <= 0) return EILEV_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool uniform = true;
    for (int r = 0; r < world; ++r) {
        if (rows[r] < 0) return EILEV_E_BADARG;
        uniform = uniform && rows[r] == rows[0];
    }
    if (rows[rank] > 0 && !local) return EILEV_E_BADARG;
    if (!comm) {  // no communicator: only a single rank can do without one
        if (world != 1) return EILEV_E_BADARG;
        if (rows[0] && local != all)
            EILEV_HIP_CHECK(hipMemcpyAsync(all, local, (size_t)(rows[0] * row_bytes), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (!g_rccl.handle) return EILEV_E_UNSUPPORTED;
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    if (uniform) {
        if (rows[0] == 0) return 0;
        return rc(g_rccl.AllGather(local, all, (size_t)(rows[0] * row_bytes), ncclInt8, c, st));
    }
    // ragged: every rank sends its block to every peer
    int e = rc(g_rccl.GroupStart());
    if (e) return e;
    int64_t off = 0;
    for (int r = 0; r < world && !e; ++r) {
        char *dst = static_cast<char *>(all) + off * row_bytes;
        if (r == rank) {
            if (rows[r] && local != dst) {
                hipError_t he = hipMemcpyAsync(dst, local, (size_t)(rows[r] * row_bytes), hipMemcpyDeviceToDevice, st);
                if (he != hipSuccess) e = (int)he;
            }
        } else {
            if (rows[rank]) e = rc(g_rccl.Send(local, (size_t)(rows[rank] * row_bytes), ncclInt8, r, c, st));
            if (!e && rows[r]) e = rc(g_rccl.Recv(dst, (size_t)(rows[r] * row_bytes), ncclInt8, r, c, st));
        }
        off += rows[r];
    }
    const int e2 = rc(g_rccl.GroupEnd());
    return e ? e : e2;
}

// All-to-all-v over rows: rows send_rows[r] starting at row send_off[r] of `send` go to rank r; recv_rows[q] rows from rank q
// land at row recv_off[q] of `recv`.  The block a rank keeps for itself is a device copy.
int eilev_exchange_clip_tokens(void *comm, const void *send, const int64_t *send_rows, const int64_t *send_off, void *recv,
                               const int64_t *recv_rows, const int64_t *recv_off, int world, int rank, int64_t row_bytes,
                               void *stream) {
    if (!send_rows || !send_off || !recv_rows || !recv_off || world < 1 || rank < 0 || rank >= world || row_bytes <= 0)
        return EILEV_E_BADARG;
    if (send_rows[rank] != recv_rows[rank]) return EILEV_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const char *s = static_cast<const char *>(send);
    char *d = static_cast<char *>(recv);
    for (int r = 0; r < world; ++r)
        if (send_rows[r] < 0 || recv_rows[r] < 0 || (send_rows[r] && !send) || (recv_rows[r] && !recv)) return EILEV_E_BADARG;
    if (send_rows[rank])
        EILEV_HIP_CHECK(hipMemcpyAsync(d + recv_off[rank] * row_bytes, s + send_off[rank] * row_bytes,
                                       (size_t)(send_rows[rank] * row_bytes), hipMemcpyDeviceToDevice, st));
    if (world == 1) return 0;
    if (!comm) return EILEV_E_BADARG;
    if (!g_rccl.handle) return EILEV_E_UNSUPPORTED;
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    int e = rc(g_rccl.GroupStart());
    if (e) return e;
    for (int r = 0; r < world && !e; ++r) {
        if (r == rank) continue;
        if (send_rows[r]) e = rc(g_rccl.Send(s + send_off[r] * row_bytes, (size_t)(send_rows[r] * row_bytes), ncclInt8, r, c, st));
        if (!e && recv_rows[r]) e = rc(g_rccl.Recv(d + recv_off[r] * row_bytes, (size_t)(recv_rows[r] * row_bytes), ncclInt8, r, c, st));
    }
    const int e2 = rc(g_rccl.GroupEnd());
    return e ? e : e2;
}

}  // extern "C"
