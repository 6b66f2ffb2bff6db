// reorder.cpp -- see reorder.hpp
#include "reorder.hpp"

#include <chrono>

#include "common.hpp"
#include "threads.hpp"

#include <atomic>

#include <algorithm>
#include <numeric>

namespace spx {

static double rcm_now()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

namespace {

template <typename P>
struct Bfs {
    const P *ptr;
    const idx_t *adj;
    std::vector<uint32_t> stamp;   // visit stamp per vertex
    std::vector<idx_t> queue;
    uint32_t epoch = 0;

    Bfs(size_t n, const P *p, const idx_t *a) : ptr(p), adj(a), stamp(n, 0) { queue.reserve(n); }

    size_t degree(idx_t v) const { return ptr[(size_t) v + 1] - ptr[(size_t) v]; }

    // Level structure rooted at `root` over vertices not finally numbered
    // (`done`).  Returns the number of levels; `last_begin` is where the last
    // level starts inside `queue`.
    size_t levels(idx_t root, const std::vector<char> &done, size_t &last_begin)
    {
        ++epoch;
        queue.clear();
        queue.push_back(root);
        stamp[(size_t) root] = epoch;
        size_t nlev = 0, head = 0;
        last_begin = 0;
        while (head < queue.size()) {
            const size_t level_end = queue.size();
            last_begin = head;
            ++nlev;
            for (; head < level_end; ++head) {
                const idx_t v = queue[head];
                for (size_t k = (size_t) ptr[(size_t) v]; k < (size_t) ptr[(size_t) v + 1]; ++k) {
                    const idx_t w = adj[k];
                    if (done[(size_t) w] || stamp[(size_t) w] == epoch) continue;
                    stamp[(size_t) w] = epoch;
                    queue.push_back(w);
                }
            }
        }
        return nlev;
    }
};

template <typename P>
void rcm_order_impl(size_t n, const P *ptr, const idx_t *adj, std::vector<idx_t> &perm)
{
    Bfs<P> bfs(n, ptr, adj);
    std::vector<char> done(n, 0);
    std::vector<idx_t> order;   // Cuthill-McKee visiting order
    order.reserve(n);
    // components are started from their lowest-degree vertex, lowest first
    std::vector<idx_t> by_degree(n);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    std::stable_sort(by_degree.begin(), by_degree.end(),
                     [&](idx_t a, idx_t b) { return bfs.degree(a) < bfs.degree(b); });
    std::vector<idx_t> nbrs;
    for (idx_t seed : by_degree) {
        if (done[(size_t) seed]) continue;
        // pseudo-peripheral vertex (George & Liu): walk to a lowest-degree
        // vertex of the deepest level until the depth stops growing
        idx_t root = seed;
        size_t last_begin = 0;
        size_t depth = bfs.levels(root, done, last_begin);
        for (;;) {
            idx_t cand = bfs.queue[last_begin];
            for (size_t k = last_begin; k < bfs.queue.size(); ++k) {
                const idx_t v = bfs.queue[k];
                if (bfs.degree(v) < bfs.degree(cand) || (bfs.degree(v) == bfs.degree(cand) && v < cand))
                    cand = v;
            }
            if (cand == root) break;
            size_t lb = 0;
            const size_t d = bfs.levels(cand, done, lb);
            if (d <= depth) break;
            depth = d;
            root = cand;
            last_begin = lb;
        }
        // Cuthill-McKee from `root`
        size_t head = order.size();
        order.push_back(root);
        done[(size_t) root] = 1;
        while (head < order.size()) {
            const idx_t v = order[head++];
            nbrs.clear();
            for (size_t k = (size_t) ptr[(size_t) v]; k < (size_t) ptr[(size_t) v + 1]; ++k) {
                const idx_t w = adj[k];
                if (!done[(size_t) w]) {
                    done[(size_t) w] = 1;
                    nbrs.push_back(w);
                }
            }
            std::sort(nbrs.begin(), nbrs.end(), [&](idx_t a, idx_t b) {
                const size_t da = bfs.degree(a), db = bfs.degree(b);
                return da != db ? da < db : a < b;
            });
            order.insert(order.end(), nbrs.begin(), nbrs.end());
        }
    }
    perm.assign(n, 0);
    for (size_t i = 0; i < n; ++i) perm[(size_t) order[n - 1 - i]] = (idx_t) i;
}

}  // namespace

void rcm_order(size_t n, const std::vector<size_t> &ptr, const std::vector<idx_t> &adj,
               std::vector<idx_t> &perm)
{
    rcm_order_impl(n, ptr.data(), adj.data(), perm);
}

void owner_order(const std::vector<idx_t> &order_perm, const std::vector<size_t> &weight, size_t world,
                 std::vector<idx_t> &perm)
{
    const size_t n = order_perm.size();
    // position -> vertex, then cut the positions into `world` ranges of equal weight with the
    // reference's rule (SparseInternal.hpp:131-144: range i takes vertices until it holds
    // (total - taken) / (world - i))
    std::vector<idx_t> at(n);
    for (size_t v = 0; v < n; ++v) at[(size_t) order_perm[v]] = (idx_t) v;
    size_t total = 0;
    for (size_t v = 0; v < n; ++v) total += weight[v];
    std::vector<uint32_t> owner(n, 0);
    std::vector<size_t> count(world + 1, 0);
    size_t taken = 0, pos = 0;
    for (size_t g = 0; g < world; ++g) {
        const size_t limit = (total - taken) / (world - g);
        size_t mine = 0;
        while (pos < n && (g + 1 == world || mine < limit)) {
            mine += weight[(size_t) at[pos]];
            owner[(size_t) at[pos]] = (uint32_t) g;
            ++count[g + 1];
            ++pos;
        }
        taken += mine;
    }
    // inside a range the vertices keep their original order
    for (size_t g = 0; g < world; ++g) count[g + 1] += count[g];
    perm.assign(n, 0);
    std::vector<size_t> fill(count.begin(), count.end() - 1);
    for (size_t v = 0; v < n; ++v) perm[v] = (idx_t) fill[owner[v]]++;
}

void dist_reorder_csr(const idx_t *rowptr, const idx_t *colind, size_t n, bool zero_based, bool pattern_symmetric,
                      size_t world, int mode, std::vector<idx_t> &perm)
{
    if (world < 1) throw FatalError("dist reorder: bad number of processes");
    const idx_t base = zero_based ? 0 : 1;
    // (the arrays are the caller's: row pointers that start at the base and never step back, columns inside
    // the matrix -- checked before either branch walks them)
    if (n && rowptr[0] != base) throw FatalError("dist reorder: row pointers do not start at the index base");
    for (size_t v = 0; v < n; ++v)
        if (rowptr[v + 1] < rowptr[v]) throw FatalError("dist reorder: row pointers step back");
    for (idx_t k = 0; n && k < rowptr[n] - base; ++k)
        if (colind[k] < base || (size_t)(colind[k] - base) >= n) throw FatalError("dist reorder: column outside the matrix");
    std::vector<size_t> weight(n);
    for (size_t v = 0; v < n; ++v) weight[v] = (size_t)(rowptr[v + 1] - rowptr[v]);
    std::vector<idx_t> order;
    if (pattern_symmetric && zero_based) {
        // the pattern is its own adjacency (a diagonal entry is a self loop: the walks skip it)
        rcm_order_impl(n, rowptr, colind, order);
    } else {
        // A + A^T without the diagonal, duplicates removed
        std::vector<size_t> ptr(n + 1, 0);
        for (size_t r = 0; r < n; ++r)
            for (idx_t k = rowptr[r] - base; k < rowptr[r + 1] - base; ++k) {
                const size_t c = (size_t)(colind[k] - base);
                if (c >= n) throw FatalError("dist reorder: column outside the matrix");
                if (c != r) {
                    ++ptr[r + 1];
                    ++ptr[c + 1];
                }
            }
        for (size_t i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
        std::vector<idx_t> adj(ptr[n]);
        {
            std::vector<size_t> fill(ptr.begin(), ptr.end() - 1);
            for (size_t r = 0; r < n; ++r)
                for (idx_t k = rowptr[r] - base; k < rowptr[r + 1] - base; ++k) {
                    const size_t c = (size_t)(colind[k] - base);
                    if (c != r) {
                        adj[fill[r]++] = (idx_t) c;
                        adj[fill[c]++] = (idx_t) r;
                    }
                }
        }
        std::vector<size_t> uptr(n + 1, 0);
        size_t w = 0;
        for (size_t v = 0; v < n; ++v) {
            const size_t b = ptr[v], e = ptr[v + 1];
            std::sort(adj.begin() + (ptrdiff_t) b, adj.begin() + (ptrdiff_t) e);
            for (size_t k = b; k < e; ++k)
                if (k == b || adj[k] != adj[k - 1]) adj[w++] = adj[k];
            uptr[v + 1] = w;
        }
        adj.resize(w);
        rcm_order_impl(n, uptr.data(), adj.data(), order);
    }
    if (mode == SPX_DIST_REORDER_RCM_OWNER) owner_order(order, weight, world, perm);
    else perm.swap(order);
}

OwnedCsrInput *reorder_rcm(MatrixInput &in, std::vector<idx_t> &perm, int mode, size_t world)
{
    perm.clear();
    const size_t n = in.nr_rows;
    if (in.nr_rows != in.nr_cols || n == 0) {
        log_msg(LOG_WARN, "no reordering available for this matrix (not square)\n");
        return nullptr;
    }
    const double t_0 = rcm_now();
    const unsigned T = host_threads();
    // the entries, 1-based: from the input's CSR arrays where it holds them (all threads), else element by element
    std::vector<Triplet> elems;
    CsrInput *c = in.as_csr();
    if (c) {
        const idx_t base = c->zero_based_ ? 0 : 1;
        const size_t total = (size_t) (c->rowptr_[n] - base);
        elems.resize(total);
        constexpr size_t ROWS = 8192;
        parallel_for((n + ROWS - 1) / ROWS, T, [&](size_t k) {
            for (size_t r = k * ROWS; r < std::min(n, (k + 1) * ROWS); ++r)
                for (size_t j = (size_t) (c->rowptr_[r] - base); j < (size_t) (c->rowptr_[r + 1] - base); ++j)
                    elems[j] = Triplet{(idx_t) (r + 1), (idx_t) (c->colind_[j] + (c->zero_based_ ? 1 : 0)), c->values_[j]};
        });
    } else {
        elems.reserve(in.nnz);
        Triplet t;
        for (in.rewind(); in.peek(t); in.advance()) elems.push_back(t);
        in.rewind();
    }
    for (const Triplet &e : elems)
        if (e.row < 1 || (size_t) e.row > n || e.col < 1 || (size_t) e.col > n) throw FatalError("entry outside the matrix");
    constexpr size_t CHUNK = (size_t) 1 << 20;
    const size_t n_chunks = (elems.size() + CHUNK - 1) / CHUNK;

    // undirected pattern graph: both directions of every off-diagonal entry, duplicates removed
    std::vector<uint32_t> deg(n + 1, 0u);
    std::atomic<size_t> off(0);
    parallel_for(n_chunks, T, [&](size_t k) {
        size_t mine = 0;
        for (size_t i = k * CHUNK; i < std::min(elems.size(), (k + 1) * CHUNK); ++i) {
            const Triplet &e = elems[i];
            if (e.row == e.col) continue;
            __atomic_fetch_add(&deg[(size_t) e.row - 1], 1u, __ATOMIC_RELAXED);
            __atomic_fetch_add(&deg[(size_t) e.col - 1], 1u, __ATOMIC_RELAXED);
            ++mine;
        }
        off.fetch_add(mine);
    });
    if (off.load() == 0) {
        log_msg(LOG_WARN, "no reordering available for this matrix\n");
        return nullptr;
    }
    std::vector<size_t> ptr(n + 1, 0);
    for (size_t i = 0; i < n; ++i) ptr[i + 1] = ptr[i] + deg[i];
    std::vector<idx_t> adj(ptr[n]);
    {
        std::vector<uint32_t> fill(n, 0u);
        parallel_for(n_chunks, T, [&](size_t k) {
            for (size_t i = k * CHUNK; i < std::min(elems.size(), (k + 1) * CHUNK); ++i) {
                const Triplet &e = elems[i];
                if (e.row == e.col) continue;
                adj[ptr[(size_t) e.row - 1] + __atomic_fetch_add(&fill[(size_t) e.row - 1], 1u, __ATOMIC_RELAXED)] = e.col - 1;
                adj[ptr[(size_t) e.col - 1] + __atomic_fetch_add(&fill[(size_t) e.col - 1], 1u, __ATOMIC_RELAXED)] = e.row - 1;
            }
        });
    }
    // every vertex' neighbours sorted, duplicates dropped in place (all threads), then closed up
    std::vector<size_t> uptr(n + 1, 0);
    {
        constexpr size_t VERTS = 8192;
        parallel_for((n + VERTS - 1) / VERTS, T, [&](size_t k) {
            for (size_t v = k * VERTS; v < std::min(n, (k + 1) * VERTS); ++v) {
                const size_t b = ptr[v], e = ptr[v + 1];
                std::sort(adj.begin() + (ptrdiff_t) b, adj.begin() + (ptrdiff_t) e);
                deg[v] = (uint32_t) (std::unique(adj.begin() + (ptrdiff_t) b, adj.begin() + (ptrdiff_t) e) - (adj.begin() + (ptrdiff_t) b));
            }
        });
        size_t w = 0;
        for (size_t v = 0; v < n; ++v) {
            const size_t b = ptr[v];
            if (w != b) std::copy(adj.begin() + (ptrdiff_t) b, adj.begin() + (ptrdiff_t) (b + deg[v]), adj.begin() + (ptrdiff_t) w);
            w += deg[v];
            uptr[v + 1] = w;
        }
        adj.resize(w);
    }

    log_msg(LOG_INFO, "Reordering input matrix...\n");
    const double t_graph = rcm_now();
    rcm_order(n, uptr, adj, perm);
    const double t_order = rcm_now();
    std::vector<idx_t>().swap(adj);
    if (mode == SPX_DIST_REORDER_RCM_OWNER && world > 1) {
        // the order only deals the rows to the processes; inside a process they keep their places
        std::vector<size_t> weight(n, 0);
        for (const Triplet &e : elems) ++weight[(size_t) e.row - 1];
        std::vector<idx_t> own;
        owner_order(perm, weight, world, own);
        perm.swap(own);
    }
    // P A P^T: the coordinates renumbered, then sorted into CSR arrays by rows (a counting sort, all threads)
    std::vector<size_t> bw_before(n_chunks, 0), bw_after(n_chunks, 0);
    parallel_for(n_chunks, T, [&](size_t k) {
        size_t b0 = 0, b1 = 0;
        for (size_t i = k * CHUNK; i < std::min(elems.size(), (k + 1) * CHUNK); ++i) {
            Triplet &e = elems[i];
            b0 = std::max<size_t>(b0, (size_t) std::abs((long) e.row - (long) e.col));
            e.row = perm[(size_t) e.row - 1] + 1;
            e.col = perm[(size_t) e.col - 1] + 1;
            b1 = std::max<size_t>(b1, (size_t) std::abs((long) e.row - (long) e.col));
        }
        bw_before[k] = b0;
        bw_after[k] = b1;
    });
    const size_t bw0 = *std::max_element(bw_before.begin(), bw_before.end()), bw1 = *std::max_element(bw_after.begin(), bw_after.end());
    std::unique_ptr<OwnedCsrInput> out(new OwnedCsrInput);
    out->nr_rows = in.nr_rows;
    out->nr_cols = in.nr_cols;
    std::vector<TripletSpan> spans;
    for (size_t k = 0; k < n_chunks; ++k) spans.push_back(TripletSpan{elems.data() + k * CHUNK, std::min(CHUNK, elems.size() - k * CHUNK)});
    csr_from_triplets(spans, n, n, false, false, out->rowptr, out->colind, out->values, T);
    out->adopt();
    log_msg(LOG_INFO, "Original Bandwidth: %zu\nFinal Bandwidth: %zu\nReordering complete\n", bw0, bw1);
    log_msg(LOG_INFO, "reordering: pattern graph %.2f s, Cuthill-McKee order %.2f s, matrix permuted and sorted %.2f s\n",
            t_graph - t_0, t_order - t_graph, rcm_now() - t_order);
    return out.release();
}

size_t bandwidth(MatrixInput &in)
{
    size_t bw = 0;
    Triplet t;
    for (in.rewind(); in.peek(t); in.advance())
        bw = std::max<size_t>(bw, (size_t) std::abs((long) t.row - (long) t.col));
    in.rewind();
    return bw;
}

}  // namespace spx
