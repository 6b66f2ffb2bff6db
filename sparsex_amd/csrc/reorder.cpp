// reorder.cpp -- see reorder.hpp
#include "reorder.hpp"

#include "common.hpp"

#include <algorithm>
#include <numeric>

namespace spx {

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

TripletInput *reorder_rcm(MatrixInput &in, std::vector<idx_t> &perm, int mode, size_t world)
{
    perm.clear();
    const size_t n = in.nr_rows;
    if (in.nr_rows != in.nr_cols || n == 0) {
        log_msg(LOG_WARN, "no reordering available for this matrix (not square)\n");
        return nullptr;
    }
    std::unique_ptr<TripletInput> out(new TripletInput);
    out->nr_rows = in.nr_rows;
    out->nr_cols = in.nr_cols;
    out->elems.reserve(in.nnz);
    Triplet t;
    for (in.rewind(); in.peek(t); in.advance()) out->elems.push_back(t);
    in.rewind();
    out->nnz = out->elems.size();

    // undirected pattern graph: both directions of every off-diagonal entry,
    // duplicates removed
    std::vector<size_t> ptr(n + 1, 0);
    size_t off = 0;
    for (const Triplet &e : out->elems)
        if (e.row != e.col) {
            ++ptr[(size_t) e.row];
            ++ptr[(size_t) e.col];
            ++off;
        }
    if (off == 0) {
        log_msg(LOG_WARN, "no reordering available for this matrix\n");
        return nullptr;
    }
    for (size_t i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    std::vector<idx_t> adj(ptr[n]);
    {
        std::vector<size_t> fill(ptr.begin(), ptr.end() - 1);
        for (const Triplet &e : out->elems)
            if (e.row != e.col) {
                adj[fill[(size_t) e.row - 1]++] = e.col - 1;
                adj[fill[(size_t) e.col - 1]++] = e.row - 1;
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

    log_msg(LOG_INFO, "Reordering input matrix...\n");
    size_t bw0 = 0, bw1 = 0;
    rcm_order(n, uptr, adj, perm);
    if (mode == SPX_DIST_REORDER_RCM_OWNER && world > 1) {
        // the order only deals the rows to the processes; inside a process they keep their places
        std::vector<size_t> weight(n, 0);
        for (const Triplet &e : out->elems) ++weight[(size_t) e.row - 1];
        std::vector<idx_t> own;
        owner_order(perm, weight, world, own);
        perm.swap(own);
    }
    for (Triplet &e : out->elems) {
        bw0 = std::max<size_t>(bw0, (size_t) std::abs((long) e.row - (long) e.col));
        e.row = perm[(size_t) e.row - 1] + 1;
        e.col = perm[(size_t) e.col - 1] + 1;
        bw1 = std::max<size_t>(bw1, (size_t) std::abs((long) e.row - (long) e.col));
    }
    std::sort(out->elems.begin(), out->elems.end(), [](const Triplet &a, const Triplet &b) {
        return a.row != b.row ? a.row < b.row : a.col < b.col;
    });
    log_msg(LOG_INFO, "Original Bandwidth: %zu\nFinal Bandwidth: %zu\nReordering complete\n", bw0, bw1);
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
