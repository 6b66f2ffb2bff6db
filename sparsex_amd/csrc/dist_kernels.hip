// dist_kernels.hip -- device side of the exchange plan of a row-partitioned
// matrix (dist.hpp) and the built-in transport: RCCL point-to-point over xGMI.
//
// A symmetric process adds into rows in front of its own (the reference's local
// buffers, src/api/matvec.c:302-318); only those entries travel, packed, to
// their owners, which add them in a fixed order (the reference's map reduction,
// src/internals/Vector.cpp:291-299).  xGMI is point-to-point: every pair of
// GPUs has its own link, so the direct pairwise exchange below is the collective
// that fits it -- a ring would push every byte over up to seven links.
#include "dist.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace spx {

#define HIP_CHECK(expr)                                                         \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            std::string m_ = std::string("HIP failure: ") + #expr + ": " +      \
                             hipGetErrorString(e_);                             \
            log_msg(LOG_ERR, "%s\n", m_.c_str());                               \
            throw FatalError(m_);                                               \
        }                                                                       \
    } while (0)

struct DistDevice {
    int device = 0;
    size_t n_send = 0, n_recv = 0, n_fix = 0;
    idx_t *send_rows = nullptr;
    double *sendbuf = nullptr, *recvbuf = nullptr;
    idx_t *fix_rows = nullptr;
    uint32_t *fix_ptr = nullptr, *fix_pos = nullptr;
    // halo of x
    size_t n_halo_send = 0, n_halo_recv = 0;
    idx_t *halo_send_rows = nullptr, *halo_cols = nullptr;
    double *halo_sendbuf = nullptr, *halo_recvbuf = nullptr;
    // overlapped step
    uint32_t *rd_pack_pos = nullptr, *rd_scat_pos = nullptr;
    std::vector<size_t> rd_pack_ptr, rd_scat_ptr;
    hipStream_t comm = nullptr;
    std::vector<hipEvent_t> ev_part;      // one per part of the product
    hipEvent_t ev_done = nullptr;
};

// sendbuf[k] = y[send_rows[k]]: the sums this process formed for rows of others
__global__ void dist_pack_kernel(const idx_t *rows, const double *y, double *buf, size_t n)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) buf[k] = y[rows[k]];
}

// y[row] += what the other processes sent for it, in the order of the senders
__global__ void dist_unpack_kernel(const idx_t *rows, const uint32_t *ptr, const uint32_t *pos,
                                   const double *buf, double *y, size_t n)
{
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double s = 0.0;
    for (uint32_t k = ptr[t]; k < ptr[t + 1]; ++k) s += buf[pos[k]];
    y[rows[t]] += s;
}

// y[cols[k]] = what the owner of entry cols[k] sent for it (distinct entries, plain stores)
__global__ void dist_scatter_kernel(const idx_t *cols, const double *buf, double *y, size_t n)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) y[cols[k]] = buf[k];
}

// the same through position lists (one round of the overlapped step)
__global__ void dist_pack_pos_kernel(const uint32_t *pos, const idx_t *rows, const double *y, double *buf, size_t n)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) buf[pos[k]] = y[rows[pos[k]]];
}
__global__ void dist_scatter_pos_kernel(const uint32_t *pos, const idx_t *cols, const double *buf, double *y, size_t n)
{
    const size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) y[cols[pos[k]]] = buf[pos[k]];
}

template <typename T>
static T *to_device(const std::vector<T> &v, size_t min_elems = 1)
{
    T *d = nullptr;
    const size_t n = std::max(v.size(), min_elems);
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(T)));
    if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

DistDevice *dist_device_create(const DistPlan &p)
{
    DistDevice *d = new DistDevice;
    HIP_CHECK(hipGetDevice(&d->device));
    d->n_send = p.send_rows.size();
    d->n_recv = p.n_recv;
    d->n_fix = p.fix_rows.size();
    d->send_rows = to_device(p.send_rows);
    d->fix_rows = to_device(p.fix_rows);
    d->fix_ptr = to_device(p.fix_ptr, 2);
    d->fix_pos = to_device(p.fix_pos);
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d->sendbuf), std::max<size_t>(d->n_send, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d->recvbuf), std::max<size_t>(d->n_recv, 1) * sizeof(double)));
    d->n_halo_send = p.halo_send_rows.size();
    d->n_halo_recv = p.halo_cols.size();
    d->halo_send_rows = to_device(p.halo_send_rows);
    d->halo_cols = to_device(p.halo_cols);
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d->halo_sendbuf), std::max<size_t>(d->n_halo_send, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d->halo_recvbuf), std::max<size_t>(d->n_halo_recv, 1) * sizeof(double)));
    return d;
}

void dist_device_free(DistDevice *d)
{
    if (!d) return;
    (void) hipFree(d->send_rows); (void) hipFree(d->fix_rows); (void) hipFree(d->fix_ptr);
    (void) hipFree(d->fix_pos); (void) hipFree(d->sendbuf); (void) hipFree(d->recvbuf);
    (void) hipFree(d->halo_send_rows); (void) hipFree(d->halo_cols);
    (void) hipFree(d->halo_sendbuf); (void) hipFree(d->halo_recvbuf);
    (void) hipFree(d->rd_pack_pos); (void) hipFree(d->rd_scat_pos);
    for (hipEvent_t e : d->ev_part) (void) hipEventDestroy(e);
    if (d->ev_done) (void) hipEventDestroy(d->ev_done);
    if (d->comm) (void) hipStreamDestroy(d->comm);
    delete d;
}

const double *dist_device_pack(DistDevice *d, const double *d_y, void *stream)
{
    if (d->n_send)
        hipLaunchKernelGGL(dist_pack_kernel, dim3((unsigned) ((d->n_send + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d->send_rows, d_y, d->sendbuf, d->n_send);
    return d->sendbuf;
}

double *dist_device_recvbuf(DistDevice *d) { return d->recvbuf; }

void dist_device_unpack(DistDevice *d, double *d_y, void *stream)
{
    if (d->n_fix)
        hipLaunchKernelGGL(dist_unpack_kernel, dim3((unsigned) ((d->n_fix + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d->fix_rows, d->fix_ptr, d->fix_pos,
                           d->recvbuf, d_y, d->n_fix);
}

const double *dist_device_halo_pack(DistDevice *d, const double *d_y, void *stream)
{
    if (d->n_halo_send)
        hipLaunchKernelGGL(dist_pack_kernel, dim3((unsigned) ((d->n_halo_send + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d->halo_send_rows, d_y, d->halo_sendbuf, d->n_halo_send);
    return d->halo_sendbuf;
}

double *dist_device_halo_recvbuf(DistDevice *d) { return d->halo_recvbuf; }

void dist_device_halo_scatter(DistDevice *d, double *d_y, void *stream)
{
    if (d->n_halo_recv)
        hipLaunchKernelGGL(dist_scatter_kernel, dim3((unsigned) ((d->n_halo_recv + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), d->halo_cols, d->halo_recvbuf, d_y, d->n_halo_recv);
}

void dist_device_set_rounds(DistDevice *d, const DistPlan &p)
{
    (void) hipFree(d->rd_pack_pos); (void) hipFree(d->rd_scat_pos);
    d->rd_pack_pos = to_device(p.rd_pack_pos);
    d->rd_scat_pos = to_device(p.rd_scat_pos);
    d->rd_pack_ptr = p.rd_pack_ptr;
    d->rd_scat_ptr = p.rd_scat_ptr;
    if (!d->comm) {
        // (highest priority: its pack / copy / scatter kernels are tiny and must not queue behind the
        // thousands of workgroups of the product part that runs next to them)
        int pri_lo = 0, pri_hi = 0;
        if (hipDeviceGetStreamPriorityRange(&pri_lo, &pri_hi) != hipSuccess) {
            (void) hipGetLastError();
            pri_hi = 0;
        }
        if (hipStreamCreateWithPriority(&d->comm, hipStreamNonBlocking, pri_hi) != hipSuccess) {
            (void) hipGetLastError();
            HIP_CHECK(hipStreamCreateWithFlags(&d->comm, hipStreamNonBlocking));
        }
        HIP_CHECK(hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
    }
    while (d->ev_part.size() < p.rounds) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->ev_part.push_back(e);
    }
}

void *dist_device_comm_stream(DistDevice *d) { return d->comm; }
double *dist_device_halo_sendbuf(DistDevice *d) { return d->halo_sendbuf; }

void dist_device_part_done(DistDevice *d, size_t part, void *main_stream)
{
    HIP_CHECK(hipEventRecord(d->ev_part.at(part), static_cast<hipStream_t>(main_stream)));
}

void dist_device_round_begin(DistDevice *d, size_t round)
{
    HIP_CHECK(hipStreamWaitEvent(d->comm, d->ev_part.at(round), 0));
}

void dist_device_round_pack(DistDevice *d, size_t r, const double *d_y)
{
    const size_t lo = d->rd_pack_ptr[r], n = d->rd_pack_ptr[r + 1] - lo;
    if (n)
        hipLaunchKernelGGL(dist_pack_pos_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d->comm,
                           d->rd_pack_pos + lo, d->halo_send_rows, d_y, d->halo_sendbuf, n);
}

void dist_device_round_scatter(DistDevice *d, size_t r, double *d_y)
{
    const size_t lo = d->rd_scat_ptr[r], n = d->rd_scat_ptr[r + 1] - lo;
    if (n)
        hipLaunchKernelGGL(dist_scatter_pos_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d->comm,
                           d->rd_scat_pos + lo, d->halo_cols, d->halo_recvbuf, d_y, n);
}

void dist_device_rounds_end(DistDevice *d, void *main_stream)
{
    HIP_CHECK(hipEventRecord(d->ev_done, d->comm));
    HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(main_stream), d->ev_done, 0));
}

// ---- RCCL transport --------------------------------------------------------------------------

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // (optional)
};

Rccl &rccl()
{
    static Rccl r;
    if (r.lib) return r;
    // loaded on demand: a single-GPU user of this library never maps librccl
    // (a copy the process has mapped already -- e.g. the one PyTorch ships -- is
    // taken first, so that there is one RCCL per process)
    for (int flags : {RTLD_NOW | RTLD_NOLOAD, RTLD_NOW | RTLD_GLOBAL}) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, flags);
            if (r.lib) break;
        }
        if (r.lib) break;
    }
    if (!r.lib) throw FatalError(std::string("cannot load librccl: ") + dlerror());
#define SPX_SYM(field, sym)                                                                      \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym));                            \
    if (!r.field) throw FatalError(std::string("librccl lacks ") + sym)
    SPX_SYM(GetUniqueId, "ncclGetUniqueId");
    SPX_SYM(CommInitRank, "ncclCommInitRank");
    SPX_SYM(CommDestroy, "ncclCommDestroy");
    SPX_SYM(GroupStart, "ncclGroupStart");
    SPX_SYM(GroupEnd, "ncclGroupEnd");
    SPX_SYM(Send, "ncclSend");
    SPX_SYM(Recv, "ncclRecv");
    SPX_SYM(GetErrorString, "ncclGetErrorString");
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
#undef SPX_SYM
    return r;
}

struct RcclCtx {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t setup_stream = nullptr;
    // set-up exchanges: a status word per peer (allocated with the communicator, so that a
    // rank can always say that something went wrong on its side) and the staging buffers
    double *st_send = nullptr, *st_recv = nullptr;       // `world` doubles each
    double *stage_s = nullptr, *stage_r = nullptr;
    size_t cap_s = 0, cap_r = 0;
};

int rccl_exchange_device(void *ctx_, const double *send, const size_t *soff, const size_t *scnt,
                         double *recv, const size_t *roff, const size_t *rcnt, void *stream)
{
    RcclCtx *c = static_cast<RcclCtx *>(ctx_);
    Rccl &r = rccl();
    hipStream_t st = static_cast<hipStream_t>(stream);
    ncclResult_t rc = r.GroupStart();
    for (int q = 0; q < c->world && rc == ncclSuccess; ++q) {
        if (q == c->rank) continue;
        if (scnt[q]) rc = r.Send(send + soff[q], scnt[q], ncclDouble, q, c->comm, st);
        if (rc == ncclSuccess && rcnt[q]) rc = r.Recv(recv + roff[q], rcnt[q], ncclDouble, q, c->comm, st);
    }
    const ncclResult_t rc2 = r.GroupEnd();
    if (rc == ncclSuccess) rc = rc2;
    if (rc != ncclSuccess) {
        log_msg(LOG_ERR, "RCCL: %s\n", r.GetErrorString(rc));
        return -1;
    }
    return 0;
}

static bool grow(double *&buf, size_t &cap, size_t want)
{
    if (want <= cap) return true;
    if (buf) (void) hipFree(buf);
    buf = nullptr;
    cap = 0;
    if (hipMalloc(reinterpret_cast<void **>(&buf), want * 8) != hipSuccess) {
        (void) hipGetLastError();
        return false;
    }
    cap = want;
    return true;
}

int rccl_exchange_host(void *ctx_, const uint64_t *send, const size_t *soff, const size_t *scnt,
                       uint64_t *recv, const size_t *roff, const size_t *rcnt)
{
    // set-up time only: staged through device buffers (8-byte words travel as doubles,
    // nothing looks at the bits).  Whatever can fail locally -- growing the staging
    // buffers, the upload -- happens BEFORE the group, and its outcome travels first, as a
    // status word per peer through buffers that exist since the communicator was made: a
    // rank that cannot take part in the payload exchange says so, and every rank returns
    // -1 together instead of waiting for it inside ncclRecv.
    RcclCtx *c = static_cast<RcclCtx *>(ctx_);
    const size_t W = (size_t) c->world;
    size_t ns = 0, nr = 0;
    for (int q = 0; q < c->world; ++q) {
        if (q == c->rank) continue;
        ns = std::max(ns, soff[q] + scnt[q]);
        nr = std::max(nr, roff[q] + rcnt[q]);
    }
    bool ok = grow(c->stage_s, c->cap_s, std::max<size_t>(ns, 1)) && grow(c->stage_r, c->cap_r, std::max<size_t>(nr, 1));
    if (ok && ns && hipMemcpyAsync(c->stage_s, send, ns * 8, hipMemcpyHostToDevice, c->setup_stream) != hipSuccess) {
        (void) hipGetLastError();
        ok = false;
    }
    // the status round: one word to and from every peer
    std::vector<double> st(W, ok ? 0.0 : 1.0), got(W, 0.0);
    std::vector<size_t> off(W), one(W, 1);
    for (size_t q = 0; q < W; ++q) off[q] = q;
    one[(size_t) c->rank] = 0;
    // (the status words go up with a synchronous copy; should even that fail, this rank STILL enters
    // the group -- with whatever the buffer held last -- because a rank that stays away leaves its
    // peers waiting inside ncclRecv, and RCCL has no timeout: it then fails locally, and the peers
    // are bounded by their caller's watchdog, bench.py's Watchdog for one)
    const bool staged = hipMemcpy(c->st_send, st.data(), W * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (!staged) (void) hipGetLastError();
    if (rccl_exchange_device(c, c->st_send, off.data(), one.data(), c->st_recv, off.data(), one.data(),
                             c->setup_stream) != 0 ||
        hipMemcpyAsync(got.data(), c->st_recv, W * 8, hipMemcpyDeviceToHost, c->setup_stream) != hipSuccess ||
        hipStreamSynchronize(c->setup_stream) != hipSuccess || !staged) {
        (void) hipGetLastError();
        log_msg(LOG_ERR, "RCCL transport: the status round of a set-up exchange failed\n");
        return -1;
    }
    for (size_t q = 0; q < W; ++q)
        if (q != (size_t) c->rank && got[q] != 0.0) ok = false;
    if (!ok) {
        log_msg(LOG_ERR, "RCCL transport: a rank could not stage its set-up exchange; all ranks give up\n");
        return -1;
    }
    if (rccl_exchange_device(c, c->stage_s, soff, scnt, c->stage_r, roff, rcnt, c->setup_stream) != 0 ||
        hipStreamSynchronize(c->setup_stream) != hipSuccess)
        return -1;
    // only the segments that were received are defined
    for (int q = 0; q < c->world; ++q)
        if (q != c->rank && rcnt[q] &&
            hipMemcpy(recv + roff[q], c->stage_r + roff[q], rcnt[q] * 8, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
    return 0;
}

}  // namespace

}  // namespace spx

extern "C" {

spx_error_t spx_hip_rccl_unique_id(void *id)
{
    static_assert(sizeof(ncclUniqueId) == SPX_RCCL_ID_BYTES, "RCCL id size");
    try {
        ncclUniqueId u;
        if (!id || spx::rccl().GetUniqueId(&u) != ncclSuccess) return SPX_FAILURE;
        memcpy(id, &u, sizeof(u));
    } catch (const spx::FatalError &e) {
        spx::log_msg(spx::LOG_ERR, "%s\n", e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
}

spx_hip_transport_t *spx_hip_transport_rccl(const void *id, int rank, int world)
{
    if (!id || world < 1 || rank < 0 || rank >= world) return NULL;
    try {
        spx::Rccl &r = spx::rccl();
        std::unique_ptr<spx::RcclCtx> c(new spx::RcclCtx);
        c->rank = rank;
        c->world = world;
        ncclUniqueId u;
        memcpy(&u, id, sizeof(u));
        const ncclResult_t rc = r.CommInitRank(&c->comm, world, u, rank);
        if (rc != ncclSuccess) {
            spx::log_msg(spx::LOG_ERR, "RCCL communicator: %s\n", r.GetErrorString(rc));
            return NULL;
        }
        if (hipStreamCreateWithFlags(&c->setup_stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&c->st_send), (size_t) world * 8) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&c->st_recv), (size_t) world * 8) != hipSuccess) {
            // (the communicator exists on the other ranks and destroying it here does not reach them:
            // a peer that goes on to its first exchange waits until its caller's watchdog ends it)
            (void) r.CommDestroy(c->comm);
            if (c->setup_stream) (void) hipStreamDestroy(c->setup_stream);
            (void) hipFree(c->st_send);
            (void) hipFree(c->st_recv);
            (void) hipGetLastError();
            return NULL;
        }
        spx_hip_transport_t *t = new spx_hip_transport_t;
        t->ctx = c.release();
        t->rank = rank;
        t->world = world;
        t->exchange_host = spx::rccl_exchange_host;
        t->exchange_device = spx::rccl_exchange_device;
        return t;
    } catch (const spx::FatalError &e) {
        spx::log_msg(spx::LOG_ERR, "%s\n", e.what.c_str());
        return NULL;
    }
}

int spx_hip_transport_rccl_ranks(const spx_hip_transport_t *t)
{
    if (!t || t->exchange_device != spx::rccl_exchange_device || !t->ctx) return -1;
    try {
        spx::Rccl &r = spx::rccl();
        const spx::RcclCtx *c = static_cast<const spx::RcclCtx *>(t->ctx);
        int n = -1;
        if (!r.CommCount || !c->comm || r.CommCount(c->comm, &n) != ncclSuccess) return -1;
        return n;
    } catch (...) {
        return -1;
    }
}

void spx_hip_transport_destroy(spx_hip_transport_t *t)
{
    // (only transports made by spx_hip_transport_rccl)
    if (!t) return;
    spx::RcclCtx *c = static_cast<spx::RcclCtx *>(t->ctx);
    if (c) {
        if (c->comm) (void) spx::rccl().CommDestroy(c->comm);
        if (c->setup_stream) (void) hipStreamDestroy(c->setup_stream);
        (void) hipFree(c->st_send); (void) hipFree(c->st_recv);
        (void) hipFree(c->stage_s); (void) hipFree(c->stage_r);
        delete c;
    }
    delete t;
}

}  // extern "C"
