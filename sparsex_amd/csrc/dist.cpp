// dist.cpp -- see dist.hpp.
#include "dist.hpp"
#include "handles.hpp"
#include "config.hpp"

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>

namespace spx {

namespace {

void host_exchange(const spx_hip_transport_t &t, const std::vector<uint64_t> &send,
                   const std::vector<size_t> &soff, const std::vector<size_t> &scnt,
                   std::vector<uint64_t> &recv, const std::vector<size_t> &roff,
                   const std::vector<size_t> &rcnt)
{
    if (t.world <= 1) return;
    if (!t.exchange_host) throw FatalError("transport without exchange_host");
    static const uint64_t dummy = 0;
    uint64_t rdummy = 0;
    if (t.exchange_host(t.ctx, send.empty() ? &dummy : send.data(), soff.data(), scnt.data(),
                        recv.empty() ? &rdummy : recv.data(), roff.data(), rcnt.data()) != 0)
        throw FatalError("transport: host exchange failed");
}

}  // namespace

// The plan is built collectively, and a rank that fails between two exchanges would leave
// the others waiting in theirs (with RCCL: forever).  So nothing throws between the first
// exchange and the last: a rank that finds something wrong locally keeps taking part, with
// empty lists, and says so in a status word that travels with the counts and once more at
// the very end -- every rank sees every status, and all of them fail together.
DistPlan *dist_build_plan(const spx_hip_transport_t &t, idx_t own_lo, idx_t own_hi, idx_t nrows,
                          const std::vector<idx_t> &conflict_rows, const std::vector<idx_t> &halo_cols,
                          bool on_device)
{
    if (t.world < 1 || t.rank < 0 || t.rank >= t.world) throw FatalError("transport: bad rank/world");
    std::unique_ptr<DistPlan> p(new DistPlan);
    p->transport = t;
    p->rank = t.rank;
    p->world = t.world;
    const size_t W = (size_t) t.world, me = (size_t) t.rank;
    std::string bad;                        // first local complaint ("" = fine so far)
    auto complain = [&bad](const char *what) { if (bad.empty()) bad = what; };

    // 1. everybody's rows
    {
        std::vector<uint64_t> send = {(uint64_t) own_lo, (uint64_t) own_hi}, recv(2 * W, 0);
        std::vector<size_t> soff(W, 0), scnt(W, 2), roff(W), rcnt(W, 2);
        for (size_t q = 0; q < W; ++q) roff[q] = 2 * q;
        scnt[me] = rcnt[me] = 0;
        host_exchange(t, send, soff, scnt, recv, roff, rcnt);
        recv[2 * me] = (uint64_t) own_lo;
        recv[2 * me + 1] = (uint64_t) own_hi;
        p->row_lo.resize(W);
        p->row_hi.resize(W);
        for (size_t q = 0; q < W; ++q) {
            p->row_lo[q] = (idx_t) recv[2 * q];
            p->row_hi[q] = (idx_t) recv[2 * q + 1];
        }
        // the ranges must tile [0, nrows) in rank order
        idx_t at = 0;
        for (size_t q = 0; q < W; ++q) {
            if (p->row_lo[q] != at || p->row_hi[q] < at) complain("row ranges of the processes do not tile the matrix");
            at = p->row_hi[q];
        }
        if (at != nrows) complain("row ranges of the processes do not cover the matrix");
    }

    // 2. this process' conflict rows, by owner (they are ascending)
    p->send_rows = conflict_rows;
    p->send_off.assign(W, 0);
    p->send_cnt.assign(W, 0);
    if (bad.empty()) {
        size_t k = 0;
        for (size_t q = 0; q < W; ++q) {
            p->send_off[q] = k;
            while (k < conflict_rows.size() && conflict_rows[k] < p->row_hi[q]) {
                if (q >= me) complain("conflict row not in front of the own rows");
                ++k;
            }
            p->send_cnt[q] = k - p->send_off[q];
        }
        if (k != conflict_rows.size()) complain("conflict row outside the matrix");
    }
    if (!bad.empty()) {                     // keeps taking part, with nothing to send
        p->send_rows.clear();
        p->send_off.assign(W, 0);
        p->send_cnt.assign(W, 0);
    }

    // 3. per peer: {entries I send you, my status: bit 0 = something is wrong here,
    //    bit 1 = I send something to somebody}
    p->recv_off.assign(W, 0);
    p->recv_cnt.assign(W, 0);
    bool any_bad = !bad.empty();
    {
        const uint64_t status = (bad.empty() ? 0u : 1u) | (p->send_rows.empty() ? 0u : 2u);
        std::vector<uint64_t> send(2 * W), recv(2 * W, 0);
        std::vector<size_t> off(W), two(W, 2);
        for (size_t q = 0; q < W; ++q) {
            send[2 * q] = p->send_cnt[q];
            send[2 * q + 1] = status;
            off[q] = 2 * q;
        }
        two[me] = 0;
        host_exchange(t, send, off, two, recv, off, two);
        size_t k = 0;
        p->any_exchange = !p->send_rows.empty();
        for (size_t q = 0; q < W; ++q) {
            p->recv_off[q] = k;
            p->recv_cnt[q] = q == me ? 0 : (size_t) recv[2 * q];
            k += p->recv_cnt[q];
            if (q != me) {
                any_bad = any_bad || (recv[2 * q + 1] & 1u);
                p->any_exchange = p->any_exchange || (recv[2 * q + 1] & 2u);
            }
        }
        p->n_recv = k;
    }
    // (every rank has seen every status word: all of them leave here, or none)
    if (any_bad)
        throw FatalError(bad.empty() ? "another process could not build its part of the exchange plan" : bad);

    // 4. the row lists themselves
    std::vector<uint64_t> recv_rows(p->n_recv, 0);
    {
        std::vector<uint64_t> send(p->send_rows.begin(), p->send_rows.end());
        host_exchange(t, send, p->send_off, p->send_cnt, recv_rows, p->recv_off, p->recv_cnt);
    }

    // 5. per own row: where in the receive buffer its sums arrive (fixed order)
    {
        std::vector<std::pair<idx_t, uint32_t>> pr;
        pr.reserve(p->n_recv);
        for (size_t k = 0; k < p->n_recv; ++k) {
            const idx_t r = (idx_t) recv_rows[k];
            if (r < own_lo || r >= own_hi) { complain("received a conflict row that is not owned here"); continue; }
            pr.push_back(std::make_pair(r, (uint32_t) k));
        }
        std::sort(pr.begin(), pr.end());
        p->fix_ptr.push_back(0);
        for (size_t k = 0; k < pr.size(); ++k) {
            if (k == 0 || pr[k].first != pr[k - 1].first) {
                if (k) p->fix_ptr.push_back((uint32_t) k);
                p->fix_rows.push_back(pr[k].first);
            }
            p->fix_pos.push_back(pr[k].second);
        }
        if (!pr.empty()) p->fix_ptr.push_back((uint32_t) pr.size());
    }

    // 5b. the halo of x: every process tells the owners which of their entries its rows read (its
    //     list is ascending, i.e. grouped by owner); an owner keeps, per process that asked, the
    //     rows to pack for it.  Same discipline as above: nothing throws between the exchanges.
    {
        p->halo_cols = halo_cols;
        p->halo_recv_off.assign(W, 0);
        p->halo_recv_cnt.assign(W, 0);
        size_t k = 0;
        for (size_t q = 0; q < W; ++q) {
            p->halo_recv_off[q] = k;
            while (k < p->halo_cols.size() && p->halo_cols[k] < p->row_hi[q]) {
                if (q == me) complain("halo column inside the own rows");
                ++k;
            }
            p->halo_recv_cnt[q] = k - p->halo_recv_off[q];
        }
        if (k != p->halo_cols.size()) complain("halo column outside the matrix");
        if (!bad.empty()) {
            p->halo_cols.clear();
            p->halo_recv_off.assign(W, 0);
            p->halo_recv_cnt.assign(W, 0);
        }
        // how many entries every process wants of mine
        std::vector<uint64_t> send(W), recv(W, 0);
        std::vector<size_t> off(W), one(W, 1);
        for (size_t q = 0; q < W; ++q) {
            send[q] = p->halo_recv_cnt[q];
            off[q] = q;
        }
        one[me] = 0;
        host_exchange(t, send, off, one, recv, off, one);
        p->halo_send_off.assign(W, 0);
        p->halo_send_cnt.assign(W, 0);
        size_t ns = 0;
        for (size_t q = 0; q < W; ++q) {
            p->halo_send_off[q] = ns;
            p->halo_send_cnt[q] = q == me ? 0 : (size_t) recv[q];
            ns += p->halo_send_cnt[q];
        }
        // ... and which
        std::vector<uint64_t> want(p->halo_cols.begin(), p->halo_cols.end()), asked(ns, 0);
        host_exchange(t, want, p->halo_recv_off, p->halo_recv_cnt, asked, p->halo_send_off, p->halo_send_cnt);
        p->halo_send_rows.resize(ns);
        for (size_t i = 0; i < ns; ++i) {
            const idx_t r = (idx_t) asked[i];
            if (r < own_lo || r >= own_hi) {
                complain("asked for a halo entry that is not owned here");
                p->halo_send_rows[i] = own_lo < own_hi ? own_lo : 0;
            } else {
                p->halo_send_rows[i] = r;
            }
        }
    }

    // 6. the slices of y, in place
    p->gat_send_off.assign(W, (size_t) own_lo);
    p->gat_send_cnt.assign(W, (size_t)(own_hi - own_lo));
    p->gat_recv_off.resize(W);
    p->gat_recv_cnt.resize(W);
    for (size_t q = 0; q < W; ++q) {
        p->gat_recv_off[q] = (size_t) p->row_lo[q];
        p->gat_recv_cnt[q] = (size_t)(p->row_hi[q] - p->row_lo[q]);
    }
    p->gat_send_cnt[me] = p->gat_recv_cnt[me] = 0;

    if (on_device && bad.empty()) {
        try {
            p->dev = dist_device_create(*p);
        } catch (const FatalError &e) {
            complain(e.what.c_str());
        }
    }

    // 7. did everybody get this far in one piece?  (a rank without its plan would leave
    //    the others waiting in the first SpMV's exchange)
    {
        std::vector<uint64_t> send(W, bad.empty() ? 0u : 1u), recv(W, 0);
        std::vector<size_t> off(W), one(W, 1);
        for (size_t q = 0; q < W; ++q) off[q] = q;
        one[me] = 0;
        host_exchange(t, send, off, one, recv, off, one);
        bool other = false;
        for (size_t q = 0; q < W; ++q) other = other || (q != me && recv[q]);
        if (!bad.empty() || other) {
            dist_device_free(p->dev);
            p->dev = nullptr;
            throw FatalError(bad.empty() ? "another process could not build its part of the exchange plan" : bad);
        }
    }
    return p.release();
}

void dist_plan_overlap(DistPlan *p, const std::vector<size_t> &chunk_rows)
{
    const size_t W = (size_t) p->world, me = (size_t) p->rank;
    constexpr size_t KMAX = DIST_MAX_CHUNKS;
    const spx_hip_transport_t &t = p->transport;
    // (a cut into more parts than the exchange has room for must not be shortened here: the device would
    // still launch, pack and send the parts behind the last one that was planned)
    if (chunk_rows.size() > KMAX + 1) throw FatalError("overlapped step: more than " + std::to_string(KMAX) + " parts");
    p->my_chunks = chunk_rows.size() > 1 ? chunk_rows.size() - 1 : 0;
    p->chunk_rows = p->my_chunks ? chunk_rows : std::vector<size_t>();
    // everybody's part bounds: {K, bounds[0..K]} padded to KMAX + 2 words
    const size_t L = KMAX + 2;
    std::vector<uint64_t> send(L, 0), recv(W * L, 0);
    send[0] = p->my_chunks;
    for (size_t k = 0; k <= p->my_chunks && p->my_chunks; ++k) send[1 + k] = p->chunk_rows[k];
    std::vector<size_t> soff(W, 0), scnt(W, L), roff(W), rcnt(W, L);
    for (size_t q = 0; q < W; ++q) roff[q] = q * L;
    scnt[me] = rcnt[me] = 0;
    if (W > 1) {
        if (!t.exchange_host || t.exchange_host(t.ctx, send.data(), soff.data(), scnt.data(), recv.data(), roff.data(), rcnt.data()) != 0)
            throw FatalError("transport: host exchange failed");
    }
    std::copy(send.begin(), send.end(), recv.begin() + me * L);
    size_t R = 0;
    for (size_t q = 0; q < W; ++q) R = std::max<size_t>(R, (size_t) recv[q * L]);
    p->rounds = R;
    if (R == 0) return;               // nobody cuts its product: no overlapped step (spx.rt.dist_chunks <= 1 everywhere)
    // round of a row of process q: the part that holds it (a process without parts: round 0)
    auto round_of = [&](size_t q, idx_t row) -> size_t {
        const size_t K = (size_t) recv[q * L];
        if (K == 0) return 0;
        const uint64_t *b = recv.data() + q * L + 1;
        size_t k = (size_t)(std::upper_bound(b, b + K + 1, (uint64_t) row) - b);
        return k == 0 ? 0 : std::min(k - 1, K - 1);
    };
    p->rd_send_off.assign(R, std::vector<size_t>(W, 0));
    p->rd_send_cnt.assign(R, std::vector<size_t>(W, 0));
    p->rd_recv_off.assign(R, std::vector<size_t>(W, 0));
    p->rd_recv_cnt.assign(R, std::vector<size_t>(W, 0));
    std::vector<std::vector<uint32_t>> pack(R), scat(R);
    for (size_t q = 0; q < W; ++q) {
        // what I send q: my rows, ascending -> my rounds in order
        size_t at = p->halo_send_off[q];
        const size_t end = at + p->halo_send_cnt[q];
        for (size_t r = 0; r < R; ++r) {
            p->rd_send_off[r][q] = at;
            while (at < end && round_of(me, p->halo_send_rows[at]) <= r) pack[r].push_back((uint32_t) at++);
            p->rd_send_cnt[r][q] = at - p->rd_send_off[r][q];
        }
        // what q sends me: its rows, ascending -> its rounds in order
        at = p->halo_recv_off[q];
        const size_t rend = at + p->halo_recv_cnt[q];
        for (size_t r = 0; r < R; ++r) {
            p->rd_recv_off[r][q] = at;
            while (at < rend && round_of(q, p->halo_cols[at]) <= r) scat[r].push_back((uint32_t) at++);
            p->rd_recv_cnt[r][q] = at - p->rd_recv_off[r][q];
        }
    }
    p->rd_pack_pos.clear(); p->rd_scat_pos.clear();
    p->rd_pack_ptr.assign(1, 0); p->rd_scat_ptr.assign(1, 0);
    for (size_t r = 0; r < R; ++r) {
        p->rd_pack_pos.insert(p->rd_pack_pos.end(), pack[r].begin(), pack[r].end());
        p->rd_scat_pos.insert(p->rd_scat_pos.end(), scat[r].begin(), scat[r].end());
        p->rd_pack_ptr.push_back(p->rd_pack_pos.size());
        p->rd_scat_ptr.push_back(p->rd_scat_pos.size());
    }
    if (p->dev) dist_device_set_rounds(p->dev, *p);
}

void dist_step_overlapped(DistPlan *p, DeviceMatrix *m, double alpha, const double *d_x, double beta,
                          double *d_y, void *stream)
{
    const spx_hip_transport_t &t = p->transport;
    if (!p->rounds || !p->dev) throw FatalError("no overlapped step planned for this matrix");
    void *comm = dist_device_comm_stream(p->dev);
    // Part r + 1 is enqueued BEFORE round r's exchange is issued: a transport that blocks the host (the
    // callback transports of the tests stage through host memory) then still has the GPU computing the
    // next part while it moves the last one's halo; with RCCL both are only enqueued anyway.
    auto launch_part = [&](size_t r) {
        if (p->my_chunks == 0) {
            if (r == 0) device_spmv(m, alpha, d_x, beta, d_y, stream);
        } else if (r < p->my_chunks) {
            device_spmv_chunk(m, r, alpha, d_x, beta, d_y, stream);
        }
        dist_device_part_done(p->dev, r, stream);          // (an event per part on the launch stream)
    };
    launch_part(0);
    for (size_t r = 0; r < p->rounds; ++r) {
        if (r + 1 < p->rounds) launch_part(r + 1);
        dist_device_round_begin(p->dev, r);                // the second stream waits for part r
        dist_device_round_pack(p->dev, r, d_y);
        if (t.exchange_device(t.ctx, dist_device_halo_sendbuf(p->dev), p->rd_send_off[r].data(), p->rd_send_cnt[r].data(),
                              dist_device_halo_recvbuf(p->dev), p->rd_recv_off[r].data(), p->rd_recv_cnt[r].data(),
                              comm) != 0)
            throw FatalError("transport: halo exchange failed");
        dist_device_round_scatter(p->dev, r, d_y);
    }
    dist_device_rounds_end(p->dev, stream);
}

void dist_free_plan(DistPlan *p)
{
    if (!p) return;
    dist_device_free(p->dev);
    delete p;
}

void dist_complete(DistPlan *p, double *d_y, bool gather, bool halo, void *stream)
{
    if (p->world <= 1) return;
    const spx_hip_transport_t &t = p->transport;
    if (!t.exchange_device) throw FatalError("transport without exchange_device");
    if (p->any_exchange) {
        const double *sendbuf = dist_device_pack(p->dev, d_y, stream);
        if (t.exchange_device(t.ctx, sendbuf, p->send_off.data(), p->send_cnt.data(),
                              dist_device_recvbuf(p->dev), p->recv_off.data(), p->recv_cnt.data(),
                              stream) != 0)
            throw FatalError("transport: device exchange failed");
        dist_device_unpack(p->dev, d_y, stream);
    }
    if (gather) {
        if (t.exchange_device(t.ctx, d_y, p->gat_send_off.data(), p->gat_send_cnt.data(), d_y,
                              p->gat_recv_off.data(), p->gat_recv_cnt.data(), stream) != 0)
            throw FatalError("transport: gathering y failed");
    } else if (halo) {
        // only what the receivers' rows read: packed, pairwise, scattered to its place in y
        // (collective: a process without a halo of its own still serves the others')
        const double *sendbuf = dist_device_halo_pack(p->dev, d_y, stream);
        if (t.exchange_device(t.ctx, sendbuf, p->halo_send_off.data(), p->halo_send_cnt.data(),
                              dist_device_halo_recvbuf(p->dev), p->halo_recv_off.data(), p->halo_recv_cnt.data(),
                              stream) != 0)
            throw FatalError("transport: halo exchange failed");
        dist_device_halo_scatter(p->dev, d_y, stream);
    }
}

}  // namespace spx

// ---- C API -------------------------------------------------------------------------------------

// Nothing leaves a C entry point as an exception -- through `extern "C"` that is std::terminate, and
// spx_mat_tune alone holds tens of gigabytes at contract size.  Every entry point of this file is a function-try-
// block that ends with SPX_C_BOUNDARY: what the inner handlers of a function do not catch becomes an error
// through the handler and the function's failure value, the reference's convention for what cannot be done
// (include/sparsex/error.h:99-115; src/api/matvec.c:259-322: spx_mat_tune returns SPX_INVALID_MAT).  A failed
// allocation is SPX_ERR_MEM_ALLOC, which the default handler, like the reference's, treats as fatal (exit(1),
// src/api/error.c:64-88); a handler set by the client sees the code and the call returns its failure value.
#define SPX_C_BOUNDARY(RETURN_STATEMENT)                                                                      \
    catch (const spx::FatalError &e_) { SETERROR_1(SPX_ERR_TUNED_MAT, e_.what.c_str()); RETURN_STATEMENT }      \
    catch (const std::bad_alloc &) { SETERROR_0(SPX_ERR_MEM_ALLOC); RETURN_STATEMENT }                         \
    catch (const std::exception &e_) { SETERROR_1(SPX_ERR_TUNED_MAT, e_.what()); RETURN_STATEMENT }            \
    catch (...) { SETERROR_1(SPX_ERR_TUNED_MAT, "unknown exception"); RETURN_STATEMENT }

extern "C" {

spx_error_t spx_hip_mat_dist_attach(spx_matrix_t *A, const spx_hip_transport_t *t)
try {
    if (!A || !t) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument"); return SPX_FAILURE; }
    try {
        if (A->dist) {
            dist_free_plan(A->dist);
            A->dist = nullptr;
        }
        A->dist = dist_build_plan(*t, A->own_lo, A->own_hi, A->nrows, A->conflict_rows, A->halo_cols,
                                  A->dev != nullptr);
        // the overlapped step (general path, spx.rt.dist_chunks parts; collective as well)
        {
            std::vector<size_t> bounds;
            // (at most DIST_MAX_CHUNKS parts: what the plan exchange has room for)
            const long K = std::min<long>((long) DIST_MAX_CHUNKS, Config::instance().get_long("spx.rt.dist_chunks"));
            if (A->dev && !A->symmetric && K > 1) device_plan_chunks(A->dev, (size_t) K, bounds);
            if (!A->dev && !A->symmetric && K > 1 && A->own_hi - A->own_lo >= 2 * K) {
                // (host-only matrix: equal parts of the rows, so that the plan itself can be tested without a GPU)
                for (long k = 0; k <= K; ++k) bounds.push_back((size_t) A->own_lo + (size_t)(A->own_hi - A->own_lo) * (size_t) k / (size_t) K);
            }
            // (collective: every process takes part, also one that does not cut its own product -- a process
            // configured with one part would otherwise leave the others waiting in the exchange of the bounds)
            dist_plan_overlap(A->dist, bounds);
        }
        // the exchange takes over what the caller-side all-reduce needed: rows this
        // process neither owns nor adds to are nobody's business any more
        // (the thin mirror list stores its rows, it needs no cleared y; spilled tile sums in
        // front of the row-blocks do)
        if (A->dev && A->symmetric) {
            idx_t first = A->first_block_row;
            if (A->has_tiles && !A->conflict_rows.empty()) first = std::min(first, A->conflict_rows.front());
            device_set_init_rows(A->dev, (size_t) first);
        }
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_mat_dist_plan(const spx_matrix_t *A, spx_hip_dist_plan_t *plan)
try {
    if (!A || !plan || !A->dist) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "matrix has no exchange plan (spx_hip_mat_dist_attach)");
        return SPX_FAILURE;
    }
    const DistPlan &p = *A->dist;
    memset(plan, 0, sizeof(*plan));
    plan->rank = p.rank;
    plan->world = p.world;
    plan->row_lo = p.row_lo.data();
    plan->row_hi = p.row_hi.data();
    plan->n_send = (int64_t) p.send_rows.size();
    plan->send_rows = p.send_rows.data();
    plan->send_off = p.send_off.data();
    plan->send_cnt = p.send_cnt.data();
    plan->n_recv = (int64_t) p.n_recv;
    plan->recv_off = p.recv_off.data();
    plan->recv_cnt = p.recv_cnt.data();
    plan->n_fix_rows = (int64_t) p.fix_rows.size();
    plan->fix_rows = p.fix_rows.data();
    plan->fix_ptr = p.fix_ptr.data();
    plan->fix_pos = p.fix_pos.data();
    plan->any_exchange = p.any_exchange ? 1 : 0;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

int spx_hip_mat_dist_parts(const spx_matrix_t *A)
try {
    return (A && A->dist) ? (int) A->dist->my_chunks : 0;
} SPX_C_BOUNDARY(return 0;)

int spx_hip_mat_dist_rounds(const spx_matrix_t *A)
try {
    return (A && A->dist) ? (int) A->dist->rounds : 0;
} SPX_C_BOUNDARY(return 0;)

spx_error_t spx_hip_mat_dist_round(const spx_matrix_t *A, int round, const size_t **send_off, const size_t **send_cnt,
                                   const size_t **recv_off, const size_t **recv_cnt)
try {
    if (!A || !A->dist || round < 0 || (size_t) round >= A->dist->rounds || !send_off || !send_cnt || !recv_off || !recv_cnt) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "no such round of the overlapped step");
        return SPX_FAILURE;
    }
    *send_off = A->dist->rd_send_off[(size_t) round].data();
    *send_cnt = A->dist->rd_send_cnt[(size_t) round].data();
    *recv_off = A->dist->rd_recv_off[(size_t) round].data();
    *recv_cnt = A->dist->rd_recv_cnt[(size_t) round].data();
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_mat_dist_halo(const spx_matrix_t *A, spx_hip_dist_halo_t *halo)
try {
    if (!A || !halo || !A->dist) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "matrix has no exchange plan (spx_hip_mat_dist_attach)");
        return SPX_FAILURE;
    }
    const DistPlan &p = *A->dist;
    memset(halo, 0, sizeof(*halo));
    halo->n_recv = (int64_t) p.halo_cols.size();
    halo->recv_cols = p.halo_cols.data();
    halo->recv_off = p.halo_recv_off.data();
    halo->recv_cnt = p.halo_recv_cnt.data();
    halo->n_send = (int64_t) p.halo_send_rows.size();
    halo->send_rows = p.halo_send_rows.data();
    halo->send_off = p.halo_send_off.data();
    halo->send_cnt = p.halo_send_cnt.data();
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

}  // extern "C"
