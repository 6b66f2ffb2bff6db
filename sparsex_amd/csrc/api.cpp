// api.cpp -- the SparseX C API (include/sparsex/*.h) on top of the host
// preprocessor and the HIP executor, plus the extensions of sparsex_hip.h.
//
// Function-by-function counterpart of the reference's src/api/matvec.c,
// src/api/common.c, src/api/error.c and the facade they call
// (src/internals/Facade.cpp:30-212).  Argument checks, return values and the
// error-handler protocol follow those files; the machinery behind tune and
// matvec is this repository's own.
#include <sparsex/sparsex.h>
#include <sparsex_hip.h>

#include "config.hpp"
#include "csx_emit.hpp"
#include "device.hpp"
#include "dist.hpp"
#include "encoder.hpp"
#include "gpu_emit.hpp"
#include "input.hpp"
#include "reorder.hpp"
#include "stream_index.hpp"
#include "threads.hpp"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <atomic>
#include <sstream>
#include <thread>
#include <unistd.h>
#include <unordered_map>

using namespace spx;

#include "handles.hpp"

namespace {

enum { ALLOC_STD = 1, ALLOC_OTHER = 4,          // Vector.cpp:36-41
       ALLOC_PINNED = 8 };                      // this build: page-locked, copied to/from HBM directly
enum { VEC_MODE_INVALID = 45 };                // Vector.cpp:43-47

// spx.vec.device (default false; env SPX_VEC_DEVICE through spx_options_set_from_env): with "true", vectors the
// library allocated itself (spx_vec_create, spx_vec_create_random -- page-locked memory of the library's own;
// NEVER spx_vec_create_from_buff vectors, whose buffer is the client's in both modes) carry a version that every
// spx_vec_* mutator advances, and spx_matvec_* reuse x's copy in HBM while it stands: a relinked reference
// client's 128-loop (test/src/sparsex_test.c:161-163) or one of its examples uploads x once and pays the way
// back of y only.  `struct vector_struct` is public, so a client CAN write through v->elements behind the
// library's back: that is why this is opt-in.  A client that opts in and writes directly calls
// spx_hip_vec_touch(v) afterwards.  The version handed to the device side also includes a fingerprint of the
// contents (a few hundred samples and both ends) as a second net: a rewritten vector is seen even without the
// call, a single poked element may not be.
std::mutex g_vec_mtx;
std::unordered_map<const spx_vector_t *, uint64_t> g_vec_version;
std::atomic<uint64_t> g_version_clock(1);

void vec_track(const spx_vector_t *v)
{
    if (!Config::instance().get_bool("spx.vec.device")) return;
    std::lock_guard<std::mutex> lk(g_vec_mtx);
    g_vec_version[v] = g_version_clock++;
}

void vec_touch(const spx_vector_t *v)
{
    std::lock_guard<std::mutex> lk(g_vec_mtx);
    auto it = g_vec_version.find(v);
    if (it != g_vec_version.end()) it->second = g_version_clock++;
}

void vec_forget(const spx_vector_t *v)
{
    std::lock_guard<std::mutex> lk(g_vec_mtx);
    g_vec_version.erase(v);
}

uint64_t vec_version(const spx_vector_t *v)
{
    uint64_t ver = 0;
    {
        std::lock_guard<std::mutex> lk(g_vec_mtx);
        auto it = g_vec_version.find(v);
        if (it == g_vec_version.end()) return 0;
        ver = it->second;
    }
    // fingerprint: 509 samples spread over the vector, its first and last elements
    uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t) v->size;
    auto mix = [&h](double d) {
        uint64_t b;
        memcpy(&b, &d, sizeof(b));
        h = (h ^ b) * 0x100000001B3ull;
    };
    const size_t n = v->size;
    if (n) {
        const size_t step = n / 509 + 1;
        for (size_t i = 0; i < n; i += step) mix(v->elements[i]);
        mix(v->elements[n - 1]);
    }
    const uint64_t out = ver * 0x9E3779B97F4A7C15ull ^ h;
    return out ? out : 1;
}

// spx.vec.register (default auto): a spx_vec_create_from_buff vector is a view of the CLIENT's buffer, pageable
// memory as a rule, and every spx_matvec_* on it would copy x into page-locked staging memory and y back out of it --
// on the bench matrix that host-side copying takes longer than the transfers.  Instead a view of 32 MB or more
// has its buffer page-locked where it lies (hipHostRegister) at its first product and from then on travels like a
// vector of the library's own; spx_vec_destroy releases it.  On this platform that costs next to nothing (0.6 ms
// for 224 MB that the client has touched, tools/micro/host_register_cost.py), so that even a client that makes a
// view per call gains (9.0 -> 6.1 ms per spx_matvec_mult on the bench matrix).  Nothing depends on it for
// correctness: a buffer that cannot be locked, or was locked by someone else and released under us, goes through
// staging or through the runtime's own pageable path.
struct VecReg {
    void *ptr = nullptr;
    size_t bytes = 0;
    unsigned uses = 0;
    bool locked = false, ours = false, failed = false;
};
std::unordered_map<const spx_vector_t *, VecReg> g_vec_reg;
constexpr unsigned REGISTER_FROM_USE = 1;
constexpr size_t VEC_REGISTER_MIN_BYTES = (size_t) 32 << 20;

bool vec_page_locked(const spx_vector_t *v)
{
    if (v->alloc_type == ALLOC_PINNED) return true;
    const size_t bytes = v->size * sizeof(spx_value_t);
    // 32 MB and more only, whatever the tests' thresholds: an allocation of that size is a mapping of its own (glibc's
    // mmap threshold never grows beyond it), so that the pages that get locked hold nothing else of the process.
    // Smaller arrays live on the heap among other objects -- among them the sources of the library's own pageable
    // uploads, which the runtime pins in place for the time of a copy; with client arrays of some hundred KB
    // page-locked next to those a soak ran into GPU memory access faults on heap addresses, one in some
    // hundred matrices (profiles/r06/NOTES.md section 4: heap pages that were page-locked and released stay mapped, are
    // handed out again to the library's vectors, and a pageable hipMemcpy from those faulted); without them, and at the
    // bench matrix' size (memory that is unmapped when the client frees it), never.
    if (bytes < VEC_REGISTER_MIN_BYTES || !v->elements) return false;
    if (Config::instance().get_str("spx.vec.register") == "false") return false;
    std::lock_guard<std::mutex> lk(g_vec_mtx);
    VecReg &r = g_vec_reg[v];
    if (r.ptr != (void *) v->elements || r.bytes != bytes) {
        if (r.ours) device_host_unregister(r.ptr);
        r = VecReg();
        r.ptr = v->elements;
        r.bytes = bytes;
    }
    ++r.uses;
    if (!r.locked && !r.failed && r.uses >= REGISTER_FROM_USE) {
        const int got = device_host_register(r.ptr, r.bytes);
        r.locked = got != 0;
        r.ours = got == 1;
        r.failed = got == 0;
        log_msg(LOG_INFO, "vector view of %.0f MB: %s\n", (double) bytes / 1048576.0,
                got == 1 ? "page-locked in place" : (got == 2 ? "already page-locked" : "cannot be page-locked, staged"));
    }
    return r.locked;
}

void vec_release_lock(const spx_vector_t *v)
{
    std::lock_guard<std::mutex> lk(g_vec_mtx);
    auto it = g_vec_reg.find(v);
    if (it == g_vec_reg.end()) return;
    if (it->second.ours) device_host_unregister(it->second.ptr);
    g_vec_reg.erase(it);
}

double now_sec()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// Nothing leaves a C entry point as an exception -- through `extern "C"` that is std::terminate, and
// spx_mat_tune alone holds tens of gigabytes at contract size.  Every entry point below is a function-try-
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

// ======================================================================================
//  error.h
// ======================================================================================

static spx_errhandler_t g_handler = err_handle;

static const char *const kErrors[] = {
    "invalid argument", "file error", "loading of input matrix failed",
    "conversion to CSX failed", "vector creation failed",
    "partitioning object wasn't properly created",
    "reordering failed to produce a permutation",
    "incompatible matrix and vector dimensions", "incompatible vector dimensions",
    "matrix entry doesn't exist", "index out of bounds", "dummy", "dummy", "dummy",
    "failed to open file", "failed to read from file", "failed to write to file",
    "memory allocation failed", "memory deallocation failed"};
static const char *const kWarnings[] = {
    "no specific file given to save CSX, using default: \"csx_file\"",
    "invalid tuning option", "invalid runtime option",
    "reordering wasn't feasible on this matrix", "entry not set"};

static const char *default_message(spx_error_t code)
{
    if (code > SPX_ERR_MIN_VALUE && code < SPX_ERR_MAX_VALUE)
        return kErrors[code - SPX_ERR_MIN_VALUE - 1];
    if (code > SPX_ERR_MAX_VALUE && code < SPX_WARN_MAX_VALUE)
        return kWarnings[code - SPX_ERR_MAX_VALUE - 1];
    return NULL;
}

void err_handle(spx_error_t code, const char *sourcefile, unsigned long lineno,
                const char *function, const char *fmt, ...)
{
    // library errors are reported and returned, OS-level errors are fatal,
    // warnings are reported (reference src/api/error.c:64-88)
    int errno_saved = errno;
    if (!fmt) fmt = default_message(code);
    if (!fmt) fmt = "unknown error code";
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    size_t len = strlen(buf);
    bool is_sys = code > SPX_ERR_SYSTEM && code < SPX_ERR_MAX_VALUE;
    if (is_sys) {
        snprintf(buf + len, sizeof(buf) - len, ": %s", strerror(errno_saved));
        len = strlen(buf);
    }
    snprintf(buf + len, sizeof(buf) - len, " [\"%s\":%ld:%s()]\n", sourcefile,
             (long) lineno, function);
    if (code > SPX_ERR_MIN_VALUE && code < SPX_ERR_MAX_VALUE) {
        log_msg(LOG_ERR, "%s", buf);
        if (is_sys) exit(1);
    } else if (code > SPX_ERR_MAX_VALUE && code < SPX_WARN_MAX_VALUE) {
        log_msg(LOG_WARN, "%s", buf);
    }
}

spx_errhandler_t spx_err_get_handler() { return g_handler; }

void spx_err_set_handler(spx_errhandler_t new_handler)
try {
    g_handler = new_handler ? new_handler : err_handle;
} SPX_C_BOUNDARY(return;)

// ======================================================================================
//  common.h
// ======================================================================================

void spx_log_disable_all() { log_set_level(LOG_NONE); }
void spx_log_error_console() { log_set_file(NULL); log_set_level(LOG_ERR); }
void spx_log_warning_console() { log_set_file(NULL); log_set_level(LOG_WARN); }
void spx_log_info_console() { log_set_file(NULL); log_set_level(LOG_INFO); }
void spx_log_verbose_console() { log_set_file(NULL); log_set_level(LOG_VERB); }
void spx_log_debug_console() { log_set_file(NULL); log_set_level(LOG_DBG); }
static const char *g_logfile = "sparsex.log";
void spx_log_error_file() { log_set_file(g_logfile); log_set_level(LOG_ERR); }
void spx_log_warning_file() { log_set_file(g_logfile); log_set_level(LOG_WARN); }
void spx_log_info_file() { log_set_file(g_logfile); log_set_level(LOG_INFO); }
void spx_log_verbose_file() { log_set_file(g_logfile); log_set_level(LOG_VERB); }
void spx_log_debug_file() { log_set_file(g_logfile); log_set_level(LOG_DBG); }
void spx_log_all_console() { log_set_file(NULL); log_set_level(LOG_DBG); }
void spx_log_all_file(const char *file)
try {
    log_set_file(file ? file : g_logfile);
    log_set_level(LOG_DBG);
} SPX_C_BOUNDARY(return;)
void spx_log_set_file(const char *file) { log_set_file(file); }

void spx_init() { log_set_level(LOG_WARN); }
void spx_finalize() {}

void *malloc_internal(size_t x, const char *sourcefile, unsigned long lineno,
                      const char *function)
{
    void *ret = malloc(x);
    if (!ret) {
        err_handle(SPX_ERR_MEM_ALLOC, sourcefile, lineno, function, NULL);
        exit(1);
    }
    return ret;
}

void free_internal(void *ptr, const char *sourcefile, unsigned long lineno,
                   const char *function)
{
    if (!ptr) {
        err_handle(SPX_ERR_MEM_FREE, sourcefile, lineno, function, NULL);
        exit(1);
    }
    free(ptr);
}

// ======================================================================================
//  input
// ======================================================================================

spx_input_t *spx_input_load_csr(const spx_index_t *rowptr, const spx_index_t *colind,
                                const spx_value_t *values, spx_index_t nrows,
                                spx_index_t ncols, ...)
try {
    va_list ap;
    va_start(ap, ncols);
    spx_option_t indexing = va_arg(ap, spx_option_t);
    va_end(ap);
    // The optional flag selects 0- or 1-based arrays; anything else means
    // 0-based.  (The reference subtracts SPX_INDEX_ZERO_BASED before
    // validating, src/api/matvec.c:174-177, which makes every input 0-based;
    // the documented meaning is implemented here.)
    bool one_based = (indexing == SPX_INDEX_ONE_BASED);

    if (!check_mat_dim(nrows) || !check_mat_dim(ncols)) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix dimensions");
        return SPX_INVALID_INPUT;
    }
    if (!rowptr) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid rowptr argument");
        return SPX_INVALID_INPUT;
    }
    if (!colind) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid colind argument");
        return SPX_INVALID_INPUT;
    }
    if (!values) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid values argument");
        return SPX_INVALID_INPUT;
    }
    // (the row pointers bound everything that is read of colind and values later: they start at the index
    // base and never step back.  The reference does not look, Csr.hpp:60-70; a library that hands the arrays
    // to sixty-four host threads and a GPU must)
    {
        const spx_index_t base = one_based ? 1 : 0;
        bool good = rowptr[0] == base;
        for (spx_index_t r = 0; r < nrows && good; ++r) good = rowptr[r + 1] >= rowptr[r];
        if (!good) {
            SETERROR_1(SPX_ERR_ARG_INVALID, "invalid rowptr argument: not ascending from the index base");
            return SPX_INVALID_INPUT;
        }
    }
    spx_input_t *A = new input;
    A->type = 'C';
    A->nrows = nrows;
    A->ncols = ncols;
    A->nnz = rowptr[nrows] - (one_based ? 1 : 0);
    A->mat = new CsrInput(rowptr, colind, values, nrows, ncols, !one_based);
    return A;
} SPX_C_BOUNDARY(return SPX_INVALID_INPUT;)

spx_input_t *spx_input_load_mmf(const char *filename)
try {
    if (!filename) {
        SETERROR_0(SPX_ERR_FILE);
        return SPX_INVALID_INPUT;
    }
    if (access(filename, F_OK | R_OK) == -1) {
        SETERROR_0(SPX_ERR_FILE);
        return SPX_INVALID_INPUT;
    }
    MmfInput *m = nullptr;
    try {
        m = new MmfInput(filename);
    } catch (const FatalError &) {
        SETERROR_1(SPX_ERR_INPUT_MAT, "loading matrix from MMF file failed");
        return SPX_INVALID_INPUT;
    }
    spx_input_t *A = new input;
    A->type = 'M';
    A->mat = m;
    A->nrows = (spx_index_t) m->nr_rows;
    A->ncols = (spx_index_t) m->nr_cols;
    A->nnz = (spx_index_t) m->nnz;
    return A;
} SPX_C_BOUNDARY(return SPX_INVALID_INPUT;)

spx_error_t spx_input_destroy(spx_input_t *A)
try {
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid input handle");
        return SPX_FAILURE;
    }
    delete A->mat;
    delete A;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

// ======================================================================================
//  tuning
// ======================================================================================

static void encode_partition(Partition &p, const EncoderParams &prm, const XformSeq &seq,
                             std::ostream *log)
{
    Encoder enc(&p, prm);
    if (seq.explicit_deltas) {
        enc.encode_serial(seq);
    } else {
        enc.remove_ignore(seq);
        enc.encode_all(log);
    }
}

static bool stream_has_pass(const GpuStream &s, uint8_t kind)
{
    for (const SpxRowBlock &rb : s.rbs)
        for (uint32_t k = 0; k < rb.n_pass; ++k)
            if (s.passes[(size_t) rb.pass_off + k].kind == kind) return true;
    return false;
}
static bool stream_has_symsegs(const GpuStream &s) { return stream_has_pass(s, SPX_PASS_SYMSEG); }

static bool stream_has_tiles(const GpuStream &s)
{
    for (const SpxRowBlock &rb : s.rbs)
        for (uint32_t k = 0; k < rb.n_pass; ++k) {
            const uint8_t kind = s.passes[(size_t) rb.pass_off + k].kind;
            if (kind == SPX_PASS_SYMTILE || kind == SPX_PASS_SYMSEG) return true;
        }
    return false;
}

// The values live in HBM; the host keeps the index arrays of the stream so that
// single entries can be found (spx_mat_get_entry / spx_mat_set_entry).
static void keep_index(spx_matrix_t *A, GpuStream &&gs)
{
    ValVec().swap(gs.values);
    std::vector<val_t>().swap(gs.dvalues);
    std::vector<val_t>().swap(gs.mirror_val);
    std::vector<uint32_t>().swap(gs.fix_idx);
    std::vector<uint32_t>().swap(gs.fix_ptr);
    std::vector<uint32_t>().swap(gs.spill_col);
    A->index.reset(new GpuStream(std::move(gs)));
}

// Automatic row-block size: about five row-blocks per compute unit, so that a
// small matrix runs as a single round of workgroups; `scale` is what the launch
// autotuner multiplies it with.
// row-blocks joined side by side (spx_matrix_t::rb_joined): rows, nonzeros, and the window budget that goes with them
constexpr size_t JOINED_ROWS = 1024, JOINED_ELEMS = 24576;
constexpr uint32_t JOINED_XW_BUDGET = 8192;

static size_t auto_target_elems(const spx_matrix_t *A, double scale)
{
    size_t local = 0;
    for (const Partition &p : A->parts) local += p.nnz * (A->symmetric ? 2 : 1);
    // (matrices far beyond the 256 MB Infinity Cache stream a little better
    // with the largest row-blocks: syn-nd24k x4, 114 M nonzeros, 162 -> 159 us)
    size_t t = std::min<size_t>(std::max<size_t>(local / 1280 + 1, 1024),
                                local > ((size_t) 64 << 20) ? SPX_MAX_RB_ELEMS : 4096);
    t = std::max<size_t>((size_t)(scale * (double) t), 512);
    return std::min<size_t>(t, SPX_MAX_RB_ELEMS);
}

// Builds the row-block descriptor stream from the encoded partitions and puts
// it into HBM (or keeps it on the host for host-only matrices).
static void emit_and_upload(spx_matrix_t *A)
{
    const size_t nown = A->parts.size();
    const size_t first = A->first_part;
    const bool sym = A->symmetric != 0;
    GpuEmitParams gp = A->emit_params;
    if (A->auto_rb) gp.target_elems = auto_target_elems(A, A->rb_scale);
    if (A->rb_joined) {
        // (general path, the launch tuner's choice for a matrix beyond the Infinity Cache: planned row-blocks
        // joined side by side -- one y tile, one set of unit windows for three times the values)
        gp.target_elems = JOINED_ELEMS;
        gp.max_rows = std::max<size_t>(gp.max_rows, JOINED_ROWS);
    }
    GpuStream gs;
    const unsigned hw = host_threads();
    // (what is to be handed back once the stream is uploaded; run on the spot if the function is left early)
    struct Deferred {
        std::function<void()> f;
        ~Deferred() { if (f) f(); }
    } deferred;
    std::function<void()> &release_after_upload = deferred.f;
    const double t_emit0 = now_sec();
    // pieces (partitions, row ranges) are emitted concurrently into streams of
    // their own and joined in order; threads left over work inside a piece
    auto emit_pieces = [&](std::vector<Partition> &pieces, const std::vector<std::vector<SymTile>> *tl,
                           const std::vector<SymSegVec> *sl = nullptr) {
        const size_t n = pieces.size();
        std::vector<GpuStream> locs(n);
        const unsigned inner = (unsigned) std::max<size_t>(1, hw / std::max<size_t>(1, std::min<size_t>(n, hw)));
        const double t_pieces = now_sec();
        parallel_for(n, hw, [&](size_t i) {
            GpuEmitParams g = gp;
            if (tl) g.tiles = &(*tl)[i];
            if (sl) g.symsegs = &(*sl)[i];
            emit_gpu(pieces[i], g, locs[i], inner);
        });
        const double t_room = now_sec();
        // The joined arrays are sized once (growing a 6 GB vector piece by piece copies it again and again), and the
        // values -- nearly all of the bytes -- are copied to their places by all host threads at once: the array is
        // sized without being written (big_alloc.hpp), so that its pages are first touched by the copying threads.
        // Offsets as append_stream would choose them: every non-empty piece starts on an even element.
        std::vector<uint64_t> v_at(n, UINT64_MAX);
        {
            size_t nv = gs.values.size(), nd = gs.descs.size(), np = gs.passes.size(), nc = gs.cidx.size(),
                   ns = gs.segrows.size(), nr = gs.rbs.size();
            for (size_t i = 0; i < n; ++i) {
                const GpuStream &l = locs[i];
                if (!(l.rbs.empty() && l.shared.empty())) {
                    nv += nv % 2;
                    v_at[i] = nv;
                    nv += l.values.size();
                }
                nd += l.descs.size();
                np += l.passes.size();
                nc += l.cidx.size() + 16;
                ns += l.segrows.size();
                nr += l.rbs.size();
            }
            const size_t had = gs.values.size();
            gs.values.resize(nv);
            gs.descs.reserve(nd);
            gs.passes.reserve(np);
            gs.cidx.reserve(nc);
            gs.segrows.reserve(ns);
            gs.rbs.reserve(nr);
            val_t *base = gs.values.data();
            struct Copy { const val_t *src; size_t at, len; };
            std::vector<Copy> copies;
            const size_t chunk = (size_t) 4 << 20;          // 32 MB of values per task
            size_t end = had;
            for (size_t i = 0; i < n; ++i) {
                if (v_at[i] == UINT64_MAX) continue;
                for (size_t k = end; k < v_at[i]; ++k) base[k] = 0.0;       // (the padding element in between)
                const size_t len = locs[i].values.size();
                for (size_t o = 0; o < len; o += chunk)
                    copies.push_back(Copy{locs[i].values.data() + o, (size_t) v_at[i] + o, std::min(chunk, len - o)});
                end = v_at[i] + len;
            }
            parallel_for(copies.size(), hw, [&](size_t k) {
                std::memcpy(base + copies[k].at, copies[k].src, copies[k].len * sizeof(val_t));
            });
            // (and the pieces' own copies go, on all threads as well)
            parallel_for(n, hw, [&](size_t i) {
                if (v_at[i] != UINT64_MAX) ValVec().swap(locs[i].values);
            });
        }
        const double t_join = now_sec();
        for (size_t i = 0; i < n; ++i) append_stream(gs, std::move(locs[i]), v_at[i]);
        log_msg(LOG_INFO, "descriptor stream: %zu pieces emitted in %.2f s, values placed in %.2f s, index arrays joined in %.2f s\n", n,
                t_room - t_pieces, t_join - t_room, now_sec() - t_join);
    };
    if (sym) {
        // The GPU stream holds the stored lower triangle and its mirror image
        // as one general matrix over rows [0, last owned row): every row is
        // then owned by exactly one row-block of this process and no atomics
        // are needed; the diagonal goes through csx_sym_init_kernel.
        gs.dvalues.assign((size_t) A->nrows, 0.0);
        for (size_t i = 0; i < nown; ++i) {
            const PartBounds &b = A->bounds[first + i];
            for (size_t r = 0; r < A->diag[i].size() && r < (size_t) b.nr_rows; ++r)
                gs.dvalues[(size_t) b.row_start + r] = A->diag[i][r];
        }
        // One process holding every partition writes each row exactly once:
        // the diagonal term and beta*y go into the kernel's write-out.  A
        // process with a slice produces a partial vector instead (rows it
        // does not touch are zeroed by the init kernel) to be summed later.
        gs.sym_fused = A->own_lo == 0 && A->own_hi == A->nrows;
        gp.skip_empty = !gs.sym_fused;
        if (gp.sym_once) {
            // the dense 8x8 tiles of the lower triangle are read once and used
            // twice (SPX_PASS_SYMTILE); everything else is held with its mirror
            // image.  One piece per owned partition, plus -- for a process with
            // a slice -- pieces for the rows in front of it that its mirror
            // image reaches.
            std::vector<SymRange> ranges;
            if (A->own_lo > 0) {
                const idx_t k = (idx_t) std::max<size_t>(1, std::min<size_t>(hw, (size_t) A->own_lo / 8192));
                for (idx_t q = 0; q < k; ++q) {
                    const idx_t lo = (idx_t)((int64_t) A->own_lo * q / k) & ~7;
                    const idx_t hi = q + 1 == k ? A->own_lo : ((idx_t)((int64_t) A->own_lo * (q + 1) / k) & ~7);
                    if (hi > lo) ranges.push_back(SymRange{lo, hi});
                }
            }
            for (size_t i = 0; i < nown; ++i) {
                const PartBounds &b = A->bounds[first + i];
                const idx_t hi = i + 1 == nown ? A->own_hi : A->bounds[first + i + 1].row_start;
                ranges.push_back(SymRange{b.row_start, hi});
            }
            std::vector<Partition> fulls;
            std::vector<std::vector<SymTile>> tiles;
            std::vector<MirrorPoint> thin;
            // read-once row segments need the atomic hand-over (their fall-back adds straight
            // to y) and so exclude the deterministic mode
            std::vector<SymSegVec> segs;
            size_t n_seg_elems = 0, n_lower = 0;
            for (const Partition &pt : A->parts) n_lower += pt.nnz;
            const size_t min_lower = (size_t) 16 << 20;       // (auto: see below)
            const bool want_segs = gp.sym_segments != 0 && !A->deterministic && A->spill_mode != 0 && !A->wave_tiles &&
                                   (gp.sym_segments == 1 || n_lower >= min_lower);
            build_sym_ranges(A->parts, ranges, gp.max_rows >= 8, fulls, tiles, hw,
                             gs.sym_fused ? nullptr : &thin, want_segs ? &segs : nullptr, gp.sym_min_run, gp.sym_max_run);
            for (const auto &v : segs)
                for (const SymSeg &sg : v) n_seg_elems += sg.width;
            // (auto: worth it where most of the stored triangle lies in such runs and the
            // matrix does not stay in the Infinity Cache anyway -- measured on syn-nlpkkt: 181 MB
            // of values 6 % slower, 433 MB 24 % faster, 5.9 GB 15 % faster than with the mirror
            // image stored; syn-cant, 30 MB, 25 % slower)
            const bool use_segs = want_segs && n_seg_elems > 0 &&
                                  (gp.sym_segments == 1 || 2 * n_seg_elems >= n_lower);
            if (want_segs && !use_segs) {
                // not worth it: the segments go back to the mirrored path
                segs.clear();
                fulls.clear();
                tiles.clear();
                thin.clear();
                build_sym_ranges(A->parts, ranges, gp.max_rows >= 8, fulls, tiles, hw,
                                 gs.sym_fused ? nullptr : &thin, nullptr);
            }
            A->has_symsegs = use_segs;
            if (use_segs) A->sym_atomic = true;
            // (read-once segments always hand over atomically, on top of the init pass: rows
            // without nonzeros of their own -- the state rows of a KKT system, whose stored
            // triangle lies in the multiplier rows -- need no row-block)
            if (use_segs) gp.skip_empty = true;
            emit_pieces(fulls, &tiles, use_segs ? &segs : nullptr);
            // (handed back by a thread of the matrix handle, like the encoded partitions further down)
            {
                struct Gone {
                    std::vector<Partition> fulls;
                    std::vector<SymSegVec> segs;
                    std::vector<std::vector<SymTile>> tiles;
                };
                Gone *g = new Gone();
                g->fulls.swap(fulls);
                g->segs.swap(segs);
                g->tiles.swap(tiles);
                // (started once the stream is in HBM: sixteen threads handing pages back would be in the way of the
                // finalisation below)
                release_after_upload = [g] {
                    parallel_for(g->fulls.size(), host_threads(), [&](size_t i) {
                        g->fulls[i] = Partition();
                        if (i < g->segs.size()) SymSegVec().swap(g->segs[i]);
                        if (i < g->tiles.size()) std::vector<SymTile>().swap(g->tiles[i]);
                    });
                    delete g;
                };
            }
            // thinly spread mirror image on rows of other processes: a CSR over those rows
            for (size_t k = 0; k < thin.size(); ++k) {
                if (k == 0 || thin[k].row != thin[k - 1].row) {
                    gs.mirror_rows.push_back((uint32_t) thin[k].row);
                    gs.mirror_ptr.push_back((uint32_t) k);
                }
                gs.mirror_col.push_back((uint32_t) thin[k].col);
                gs.mirror_val.push_back(thin[k].val);
            }
            gs.mirror_ptr.push_back((uint32_t) thin.size());
            gs.nnz_stored += thin.size();
            gs.n_delta_elems += thin.size();
        } else {
            Partition full;
            for (size_t i = 0; i < nown; ++i) append_sym_expanded(A->parts[i], full, gp.sym_remine);
            if (gs.sym_fused) full.nr_rows = (size_t) A->nrows;
            emit_gpu(full, gp, gs, hw);
        }
    } else if (A->col_phases <= 1) {
        emit_pieces(A->parts, nullptr);
    } else {
        // column phases: the matrix as a sum of column slices, every slice a run of row-blocks
        // of its own that is launched after the one in front of it (SPX_RB_PHASE_START).  A unit
        // goes where its anchor column lies.  Slice 0 covers every row (it stores y), the
        // others only rows that hold something of theirs (they add).
        const size_t K = A->col_phases;
        GpuEmitParams keep = gp;
        gp.max_rows = std::max<size_t>(gp.max_rows, SPX_MAX_WIDE_ROWS);    // (a slice of a row is short)
        for (size_t k = 0; k < K; ++k) {
            const idx_t c_lo = (idx_t)((int64_t) A->ncols * (int64_t) k / (int64_t) K) + 1;           // 1-based
            const idx_t c_hi = (idx_t)((int64_t) A->ncols * (int64_t)(k + 1) / (int64_t) K) + 1;
            std::vector<Partition> sub(nown);
            parallel_for(nown, hw, [&](size_t i) {
                const Partition &p = A->parts[i];
                Partition &q = sub[i];
                q.nr_rows = p.nr_rows; q.nr_cols = p.nr_cols; q.type = p.type; q.row_start = p.row_start;
                q.pool = p.pool;
                for (size_t e = 0; e < p.elems_size; ++e)
                    if (p.elems[e].col >= c_lo && p.elems[e].col < c_hi) {
                        q.elems.push_back(p.elems[e]);
                        q.nnz += p.elems[e].size;
                    }
                q.elems_size = q.elems.size();
            });
            gp.skip_empty = k > 0 || A->col_concurrent;
            const size_t rb0 = gs.rbs.size();
            emit_pieces(sub, nullptr);
            if (k > 0 && gs.rbs.size() > rb0) gs.rbs[rb0].flags |= SPX_RB_PHASE_START;
            // (a concurrent slice without a single row-block would leave its XCDs' lists undefined)
            if (A->col_concurrent && gs.rbs.size() == rb0) throw FatalError("column phases: an empty slice");
        }
        if (A->col_concurrent)
            for (SpxRowBlock &rb : gs.rbs) rb.flags |= SPX_RB_ACCUM;
        gp = keep;
        // (an over-long row is summed by a fix-up kernel that stores: not with slices that add)
        if (!gs.shared.empty()) throw FatalError("column phases: the matrix holds rows that are split over row-blocks");
    }
    const double t_emit1 = now_sec();
    A->conflict_rows.clear();
    if (sym && !gs.sym_fused) stream_touched_rows(gs, A->own_lo, A->conflict_rows);
    A->halo_cols.clear();
    if (A->own_lo > 0 || A->own_hi < A->nrows) stream_read_cols(gs, A->own_lo, A->own_hi, (size_t) A->ncols, A->halo_cols);
    A->first_block_row = A->own_lo;
    for (const SpxRowBlock &rb : gs.rbs) A->first_block_row = std::min<idx_t>(A->first_block_row, (idx_t) rb.row0);
    finalize_stream(gs, (size_t) A->nrows);
    if (sym) mark_private_rowblocks(gs, (size_t) A->nrows, A->own_lo, A->own_hi);
    std::vector<std::pair<uint32_t, uint32_t>>().swap(gs.direct_cols);
    gs.waves = (uint32_t) A->waves;
    gs.band_order = Config::instance().get_bool("spx.gpu.band_order");
    gs.arena = Config::instance().get_bool("spx.gpu.arena");
    gs.xw_budget = (A->unit_windows != 0 && !A->deterministic && !sym) ? (A->rb_joined ? std::max(A->xw_budget, JOINED_XW_BUDGET) : A->xw_budget) : 0u;
    gs.xw_gap = A->xw_gap;
    gs.xw_on = A->xw_on;
    gs.sx_plan = sym && A->sym_pipeline != 0 && !A->deterministic && A->wave_tiles != 1;
    gs.sx_on = A->sx_on;
    gs.sym_atomic = A->sym_atomic && !A->deterministic;
    gs.deterministic = A->deterministic;
    gs.wave_tiles = A->deterministic || A->wave_tiles == 1;
    A->nnz_stored = gs.nnz_stored;
    A->n_unit_elems = gs.n_unit_elems;
    A->n_delta_elems = gs.n_delta_elems;
    A->n_units = gs.n_units;
    A->value_bytes = gs.values.size() * sizeof(val_t);
    A->index_bytes = gs.index_bytes();
    A->n_rowblocks = gs.rbs.size();
    A->n_shared = gs.shared.size();
    A->has_tiles = stream_has_tiles(gs);
    A->has_symtiles = stream_has_pass(gs, SPX_PASS_SYMTILE);
    if (A->dev) {
        device_free(A->dev);
        A->dev = nullptr;
    }
    const double t_emit2 = now_sec();
    if (!A->host_only) {
        A->dev = device_upload(gs, (size_t) A->nrows, (size_t) A->ncols, sym, A->own_lo, A->own_hi,
                               A->device_ordinal);
        log_msg(LOG_INFO, "descriptor stream: emitted in %.2f s, finalized in %.2f s, uploaded in %.2f s (%zu row-blocks)\n",
                t_emit1 - t_emit0, t_emit2 - t_emit1, now_sec() - t_emit2, gs.rbs.size());
        // (a matrix that is attached to an exchange plan keeps its limited init range over a
        // re-upload: the state lives with the matrix, not with the device copy)
        if (A->dist && sym) {
            idx_t first_init = A->first_block_row;
            if (A->has_tiles && !A->conflict_rows.empty()) first_init = std::min(first_init, A->conflict_rows.front());
            device_set_init_rows(A->dev, (size_t) first_init);
        }
        keep_index(A, std::move(gs));
    } else {
        A->host_stream.reset(new GpuStream(std::move(gs)));
    }
    if (release_after_upload) {
        std::function<void()> f;
        f.swap(release_after_upload);
        A->release_wait();               // (one release at a time)
        A->release_later(std::move(f));
    }
}

// Launch parameters are measured, not guessed: small matrices (one round of
// workgroups) like two wavefronts per workgroup and smaller row-blocks,
// leftover-heavy ones eight wavefronts, the rest the default.  A handful of
// candidates, a few hundred launches each; the fastest stays.
static void autotune_launch(spx_matrix_t *A, bool tune_waves, bool tune_spill, bool tune_wave_tiles, bool tune_xw)
{
    // launches per timing: a hundred for a product of microseconds, fewer where one launch takes a
    // millisecond (about 20 ms of launches per timing either way; four timings per variant: on the
    // contract matrix the hundred cost 1.7 s per variant for the same answer)
    // (nothing of ours may keep the host busy while launches are timed: a timing is as few as eight launches)
    A->release_wait();
    const double t_est = device_time_spmv(A->dev, 2, 3);
    const int N = (int) std::min(100.0, std::max(8.0, 0.02 / std::max(t_est, 1e-7)));
    const int W = std::max(2, N / 10);
    auto best_time = [&]() {
        A->release_wait();               // (a re-emission hands its ranges back on a thread of its own)
        double best = device_time_spmv(A->dev, W, N);
        for (int rep = 0; rep < 3; ++rep) best = std::min(best, device_time_spmv(A->dev, 0, N));
        return best;
    };
    // the unit windows of x in LDS (csx_spmv_xw_kernel) against the plain kernel, whatever else is
    // being varied: `xw` receives the state the returned time belongs to
    auto time_xw = [&](bool &xw) {
        if (!tune_xw || !device_has_xw(A->dev) || A->wave_tiles == 1) {
            // (nothing to choose: no windows planned, or a y tile per wavefront won -- device_set_xw does
            // nothing then, and timing it "both ways" would compare a kernel with itself)
            xw = device_get_xw(A->dev);
            return best_time();
        }
        device_set_xw(A->dev, false);
        const double t0 = best_time();
        device_set_xw(A->dev, true);
        const double t1 = best_time();
        xw = t1 < 0.985 * t0;
        device_set_xw(A->dev, xw);
        return std::min(t0, xw ? t1 : t0);
    };
    if (!tune_waves) {
        // (the wavefront count is pinned: only the hand-over of the tiles' sums is measured)
        if (tune_wave_tiles && !device_has_tiles(A->dev) && !(A->unit_windows == 1 && device_has_xw(A->dev))) {
            auto t_of = [&](bool on) {
                device_set_wave_tiles(A->dev, on);
                return best_time();
            };
            const double t0 = t_of(false), t1 = t_of(true);
            A->wave_tiles = t1 < 0.985 * t0 ? 1 : 0;
            device_set_wave_tiles(A->dev, A->wave_tiles == 1);
        }
        if (tune_xw && device_has_xw(A->dev) && A->wave_tiles != 1) {
            bool xw = false;
            (void) time_xw(xw);
            A->xw_on = xw;
        }
        if (!tune_spill || !device_has_spill(A->dev)) return;
        auto t_of = [&](bool atomic) {
            device_set_sym_atomic(A->dev, atomic);
            return best_time();
        };
        const double tl = t_of(false), ta = t_of(true);
        A->sym_atomic = ta < 0.985 * tl;
        device_set_sym_atomic(A->dev, A->sym_atomic);
        return;
    }
    auto time_with = [&](int waves, bool &xw) {
        device_set_waves(A->dev, waves);
        return time_xw(xw);
    };
    // default row-blocks: 4 and 8 wavefronts
    bool xw4 = false, xw8 = false;
    const double t4 = time_with(4, xw4), t8 = time_with(8, xw8);
    int best_waves = t8 < 0.985 * t4 ? 8 : 4;
    bool best_xw = best_waves == 8 ? xw8 : xw4;
    double best_t = std::min(t4, t8), best_scale = 1.0;
    device_set_xw(A->dev, best_xw);
    A->xw_on = best_xw;
    // a y tile per wavefront (no two wavefronts add to the same LDS address; more LDS,
    // a reduction before the write-out): pays where rows are spread over many passes
    // (syn-nlpkkt 291 -> 274 us), costs where the kernel is launch-bound (syn-cant 7.2 -> 7.9)
    // (not against unit windows that were asked for: the per-wavefront tiles run through the plain kernel)
    if (tune_wave_tiles && !device_has_tiles(A->dev) && !(A->unit_windows == 1 && device_has_xw(A->dev))) {
        device_set_wave_tiles(A->dev, true);
        device_set_waves(A->dev, best_waves);
        const double tw = best_time();
        if (tw < 0.985 * best_t) {
            best_t = tw;
            A->wave_tiles = 1;
            best_xw = false;              // (the per-wavefront tiles run through the plain kernel)
            A->xw_on = false;             // ... also in whatever is emitted again further down
        } else {
            device_set_wave_tiles(A->dev, false);
            A->wave_tiles = 0;
            device_set_xw(A->dev, best_xw);
        }
    }
    // symmetric tiles: the transposed sums through the spill array and a second
    // kernel, or straight into y with global atomics
    if (tune_spill && device_has_spill(A->dev)) {
        device_set_sym_atomic(A->dev, !A->sym_atomic);
        device_set_waves(A->dev, best_waves);
        const double ta = best_time();
        if (ta < 0.985 * best_t) {
            best_t = ta;
            A->sym_atomic = !A->sym_atomic;
        } else {
            device_set_sym_atomic(A->dev, A->sym_atomic);
        }
    }
    // smaller row-blocks with 2 wavefronts: only worth trying where the whole
    // matrix is in flight at once anyway (needs a second emission + upload)
    if (A->auto_rb && A->n_rowblocks <= 4096) {
        for (double scale : {0.7, 0.52}) {
            A->rb_scale = scale;
            A->waves = 2;
            emit_and_upload(A);
            bool xw2 = false;
            const double t2 = time_with(2, xw2);
            if (t2 < 0.985 * best_t) {
                best_t = t2;
                best_waves = 2;
                best_scale = scale;
                best_xw = xw2;
            }
        }
        A->xw_on = best_xw;
        if (A->rb_scale != best_scale) {
            A->rb_scale = best_scale;
            A->waves = best_waves;
            emit_and_upload(A);
        }
    }
    // many row-blocks (several rounds of workgroups): twice the size halves the
    // per-row-block overhead (syn-nlpkkt N = 60: 33.0 -> 30.4 us)
    // (not where the row-blocks already have the largest size: the second
    // emission would produce the same stream)
    if (A->auto_rb && A->n_rowblocks > 4096 &&
        auto_target_elems(A, 2.0) != auto_target_elems(A, best_scale)) {
        A->rb_scale = 2.0;
        A->waves = best_waves;
        emit_and_upload(A);
        bool xwb = false;
        const double tb = time_with(best_waves, xwb);
        if (tb < 0.985 * best_t) {
            best_t = tb;
            best_scale = 2.0;
            best_xw = xwb;
            A->xw_on = best_xw;
        } else {
            A->rb_scale = best_scale;
            A->xw_on = best_xw;
            emit_and_upload(A);
        }
    }
    A->rb_scale = best_scale;
    A->waves = best_waves;
    A->xw_on = best_xw;
    // A stream far beyond the Infinity Cache whose unit passes read x from LDS: row-blocks joined side by side,
    // eight wavefronts -- longer-lived workgroups that stage half as much x per value byte (measured -1.7 ... -2 %
    // on the bench matrix in two calls of round 5, profiles/r05/xw_budget_and_rowblock_size_raw.md; small
    // matrices lose, so only here).  One more emission; kept only if it is faster.
    if (A->auto_rb && best_xw && A->wave_tiles != 1 && A->emit_params.max_rows <= SPX_MAX_RB_ROWS &&
        A->value_bytes > ((size_t) 1 << 30)) {
        A->rb_joined = true;
        A->waves = 8;
        emit_and_upload(A);
        bool xwj = false;
        const double tj = time_with(8, xwj);
        if (xwj && tj < 0.995 * best_t) {
            best_t = tj;
            best_waves = 8;
            log_msg(LOG_INFO, "launch autotune: row-blocks joined side by side (%zu of them)\n", A->n_rowblocks);
        } else {
            A->rb_joined = false;
            A->waves = best_waves;
            A->xw_on = best_xw;
            emit_and_upload(A);
        }
    }
    A->waves = best_waves;
    A->xw_on = best_xw;
    device_set_waves(A->dev, best_waves);
    device_set_xw(A->dev, best_xw);
    log_msg(LOG_INFO, "launch autotune: %d wavefronts per workgroup, row-block scale %.2f%s, unit windows %s (%.2f us per SpMV)\n",
            best_waves, best_scale, A->rb_joined ? ", joined" : "", best_xw ? "on" : "off", 1e6 * best_t);
}

static spx_matrix_t *do_tune(spx_input_t *in)
{
    Config &cfg = Config::instance();
    const double t0 = now_sec();
    const size_t P = cfg.nr_partitions();
    long world = std::max<long>(1, cfg.get_long("spx.rt.gpu_world"));
    long rank = cfg.get_long("spx.rt.gpu_rank");
    if (rank < 0 || rank >= world || P % (size_t) world != 0) {
        log_msg(LOG_ERR, "spx.rt.gpu_rank/gpu_world (%ld/%ld) do not divide "
                "spx.rt.nr_threads=%zu partitions\n", rank, world, P);
        throw FatalError("bad gpu_rank/gpu_world");
    }
    const size_t first = (size_t) rank * (P / (size_t) world);
    const size_t last = first + P / (size_t) world;
    // A process may hand over just the rows it owns of a larger matrix
    // (spx.rt.row_offset / spx.rt.global_rows): partitions, row-blocks and the
    // vectors of spx_matvec_* are then numbered globally, exactly as when it
    // holds a slice of the partitions of a matrix given in full.
    const long slice_rows = cfg.get_long("spx.rt.global_rows");
    const long slice_off = slice_rows > 0 ? cfg.get_long("spx.rt.row_offset") : 0;
    if (slice_rows > 0 && (world != 1 || slice_off < 0 || slice_off + (long) in->nrows > slice_rows)) {
        log_msg(LOG_ERR, "spx.rt.row_offset/global_rows (%ld/%ld) do not hold the %ld input rows "
                "(and exclude spx.rt.gpu_world > 1)\n", slice_off, slice_rows, (long) in->nrows);
        throw FatalError("bad row slice");
    }
    in->mat->row_base = (size_t) slice_off;
    in->mat->global_rows = slice_rows > 0 ? (size_t) slice_rows : 0;
    const idx_t rows_end = slice_rows > 0 ? (idx_t)(slice_off + in->nrows) : in->nrows;   // last input row + 1, global
    const bool sym = cfg.get_bool("spx.matrix.symmetric");
    const bool host_only = cfg.get_bool("spx.rt.host_only");
    EncoderParams prm = EncoderParams::from_config(cfg);
    XformSeq seq = cfg.xform();
    cfg.cpu_affinity();   // validates the list
    log_msg(LOG_INFO, "Format: %s\n", sym ? "CSX-sym" : "CSX");
    if (!host_only && device_count() <= 0) {
        log_msg(LOG_ERR, "no usable HIP device: this library multiplies on an MI355X "
                "only (spx.rt.host_only=true preprocesses without one)\n");
        throw FatalError("no HIP device");
    }

    std::unique_ptr<matrix> A(new matrix);
    A->nrows = slice_rows > 0 ? (spx_index_t) slice_rows : in->nrows;
    A->ncols = in->ncols;
    A->nnz = in->nnz;
    A->symmetric = sym ? 1 : 0;
    A->permutation = SPX_INVALID_PERM;
    A->nr_partitions = P;
    A->first_part = first;
    A->last_part = last;
    A->full_colind = cfg.get_bool("spx.matrix.full_colind");
    A->dev = nullptr;
    A->host_only = host_only;
    A->device_ordinal = (int) cfg.get_long("spx.rt.device");

    const size_t nown = last - first;
    std::vector<std::ostringstream> logs(nown);
    std::vector<std::string> errors(nown);
    std::vector<PartitionSym> sparts;
    if (sym) {
        build_partitions_sym(*in->mat, P, first, last, sparts, A->bounds);
        A->diag.resize(nown);
    } else {
        build_partitions(*in->mat, P, first, last, A->parts, A->bounds);
    }

    const double t_parts = now_sec();
    log_msg(LOG_INFO, "partitions built in %.2f s\n", t_parts - t0);
    // one preprocessing thread per owned partition, as the reference does
    // (CsxBuild.hpp:290-326, :344-380)
    auto work = [&](size_t i) {
        try {
            std::ostream *lg = &logs[i];
            *lg << "==> Thread: #" << (first + i) << "\n==== ENCODING STATISTICS ====\n";
            if (sym) {
                PartitionSym &ps = sparts[i];
                ps.divide();
                // the first partition has nothing left of its own rows
                // (CsxBuild.hpp:236-244)
                Encoder e1(&ps.m1, prm), e2(&ps.m2, prm);
                if (seq.explicit_deltas) {
                    e1.encode_serial(seq);
                    e2.encode_serial(seq);
                } else {
                    e1.remove_ignore(seq);
                    e2.remove_ignore(seq);
                    if (ps.lower.row_start > 0) e1.encode_all(lg);
                    e2.encode_all(lg);
                }
                ps.merge();
            } else {
                encode_partition(A->parts[i], prm, seq, lg);
            }
        } catch (const FatalError &e) {
            errors[i] = e.what;
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < nown; ++i) th.emplace_back(work, i);
        if (nown) work(0);
        for (auto &t : th) t.join();
    }
    for (auto &e : errors)
        if (!e.empty()) throw FatalError(e);
    log_msg(LOG_INFO, "partitions mined and encoded in %.2f s\n", now_sec() - t_parts);
    if (sym) {
        A->parts.resize(nown);
        for (size_t i = 0; i < nown; ++i) {
            A->parts[i] = std::move(sparts[i].lower);
            A->diag[i] = std::move(sparts[i].diagonal);
        }
    }
    // Rows after the last stored nonzero belong to the last partition: the
    // executor has to write them (y = beta*y there).  The reference leaves
    // them to the caller-side VecInit (CsxKernels.cpp:93).
    if (last == P && nown) {
        PartBounds &b = A->bounds[P - 1];
        if (b.row_start + b.nr_rows < rows_end) {
            b.nr_rows = rows_end - b.row_start;
            A->parts[nown - 1].nr_rows = (size_t) b.nr_rows;
        }
    }
    for (size_t i = 0; i < nown; ++i) A->log += logs[i].str();
    log_msg(LOG_VERB, "%s", A->log.c_str());
    A->tune_seconds = now_sec() - t0;

    // descriptor stream + upload
    const double t1 = now_sec();
    long rbe = cfg.get_long("spx.gpu.rowblock_elems");
    A->auto_rb = rbe <= 0;
    A->emit_params.target_elems = (size_t) std::max<long>(64, rbe);
    A->emit_params.max_rows = (size_t) std::min<long>(SPX_MAX_WIDE_ROWS, std::max<long>(1, cfg.get_long("spx.gpu.rowblock_rows")));
    A->emit_params.sym_remine = cfg.get_bool("spx.gpu.sym_remine");
    A->emit_params.sym_once = cfg.get_bool("spx.gpu.sym_once");
    A->emit_params.recut_linear = cfg.get_bool("spx.gpu.recut_linear");
    A->emit_params.keep_units = cfg.get_bool("spx.gpu.keep_units");
    A->emit_params.inline_desc = cfg.get_bool("spx.gpu.inline_desc");
    A->emit_params.sym_pure_passes = cfg.get_bool("spx.gpu.sym_pure_passes");
    A->emit_params.x_window = cfg.get_bool("spx.gpu.x_window");
    {
        const std::string m = cfg.get_str("spx.gpu.sym_segments");
        if (m != "auto" && m != "true" && m != "false") {
            log_msg(LOG_ERR, "spx.gpu.sym_segments: true, false or auto\n");
            throw FatalError("bad spx.gpu.sym_segments");
        }
        A->emit_params.sym_segments = m == "auto" ? -1 : (m == "true" ? 1 : 0);
        const long wide = cfg.get_long("spx.gpu.sym_wide_rows");
        if (wide < 1 || wide > SPX_MAX_WIDE_ROWS) {
            log_msg(LOG_ERR, "spx.gpu.sym_wide_rows: 1 .. %d\n", SPX_MAX_WIDE_ROWS);
            throw FatalError("bad spx.gpu.sym_wide_rows");
        }
        A->emit_params.wide_rows = (size_t) wide;
        const long mr = cfg.get_long("spx.gpu.sym_segment_min");
        if (mr < 2 || mr > 8) {
            log_msg(LOG_ERR, "spx.gpu.sym_segment_min: 2 .. 8\n");
            throw FatalError("bad spx.gpu.sym_segment_min");
        }
        A->emit_params.sym_min_run = (size_t) mr;
        const long xr = cfg.get_long("spx.gpu.sym_segment_max");
        if (xr < mr || xr > SPX_MAX_SEG_WIDTH) {
            log_msg(LOG_ERR, "spx.gpu.sym_segment_max: spx.gpu.sym_segment_min .. %d\n", SPX_MAX_SEG_WIDTH);
            throw FatalError("bad spx.gpu.sym_segment_max");
        }
        A->emit_params.sym_max_run = (size_t) xr;
    }
    A->emit_params.stack_segments = cfg.get_bool("spx.gpu.stack_segments");
    {
        idx_t lo = nown ? A->bounds[first].row_start : 0;
        idx_t hi = nown ? A->bounds[last - 1].row_start + A->bounds[last - 1].nr_rows : 0;
        if (last == P) hi = rows_end;     // trailing empty rows belong to the last slice
        A->own_lo = lo;
        A->own_hi = hi;
    }
    A->waves = (int) cfg.get_long("spx.gpu.waves");
    const bool autotune = A->waves == 0;
    if (autotune) A->waves = 4;
    const std::string spill_mode = cfg.get_str("spx.gpu.sym_spill");
    if (spill_mode != "auto" && spill_mode != "lists" && spill_mode != "atomic") {
        log_msg(LOG_ERR, "spx.gpu.sym_spill: lists, atomic or auto\n");
        throw FatalError("bad spx.gpu.sym_spill");
    }
    A->deterministic = cfg.get_bool("spx.gpu.deterministic");
    A->sym_atomic = spill_mode == "atomic" && !A->deterministic;
    A->spill_mode = spill_mode == "lists" ? 0 : (spill_mode == "atomic" ? 1 : -1);
    const std::string wt_mode = cfg.get_str("spx.gpu.wave_tiles");
    if (wt_mode != "auto" && wt_mode != "true" && wt_mode != "false") {
        log_msg(LOG_ERR, "spx.gpu.wave_tiles: true, false or auto\n");
        throw FatalError("bad spx.gpu.wave_tiles");
    }
    A->wave_tiles = wt_mode == "true" ? 1 : 0;          // (auto: off until measured)
    const std::string xw_mode = cfg.get_str("spx.gpu.unit_windows");
    if (xw_mode != "auto" && xw_mode != "true" && xw_mode != "false") {
        log_msg(LOG_ERR, "spx.gpu.unit_windows: true, false or auto\n");
        throw FatalError("bad spx.gpu.unit_windows");
    }
    {
        const long xb = cfg.get_long("spx.gpu.unit_window_doubles"), xg = cfg.get_long("spx.gpu.unit_window_gap");
        if (xb < 0 || xb > 16384 || xg < 0 || xg > 255) {
            log_msg(LOG_ERR, "spx.gpu.unit_window_doubles: 0 .. 16384, spx.gpu.unit_window_gap: 0 .. 255\n");
            throw FatalError("bad spx.gpu.unit_window_doubles / spx.gpu.unit_window_gap");
        }
        A->xw_budget = (uint32_t) xb;
        A->xw_gap = (uint32_t) xg;
    }
    A->unit_windows = xw_mode == "auto" ? -1 : (xw_mode == "true" ? 1 : 0);
    A->xw_on = xw_mode == "true";                       // (auto: off until measured)
    const std::string sx_mode = cfg.get_str("spx.gpu.sym_pipeline");
    if (sx_mode != "auto" && sx_mode != "true" && sx_mode != "false") {
        log_msg(LOG_ERR, "spx.gpu.sym_pipeline: true, false or auto\n");
        throw FatalError("bad spx.gpu.sym_pipeline");
    }
    A->sym_pipeline = sx_mode == "auto" ? -1 : (sx_mode == "true" ? 1 : 0);
    A->sx_on = sx_mode == "true";                       // (auto: off until measured)
    const std::string ph_mode = cfg.get_str("spx.gpu.col_phases");
    const bool ph_conc = ph_mode.size() == 2 && ph_mode[0] == 'c';
    long ph_fixed = ph_mode == "auto" ? 0 : strtol(ph_mode.c_str() + (ph_conc ? 1 : 0), nullptr, 10);
    if (ph_mode != "auto" && (ph_fixed < 1 || ph_fixed > 8 || (ph_conc && ph_fixed != 2 && ph_fixed != 4 && ph_fixed != 8))) {
        log_msg(LOG_ERR, "spx.gpu.col_phases: 1 .. 8 (launched in turn), c2 | c4 | c8 (one launch, a group of XCDs each) or auto\n");
        throw FatalError("bad spx.gpu.col_phases");
    }
    A->col_phases = (!sym && ph_fixed > 1) ? (size_t) ph_fixed : 1;
    A->col_concurrent = A->col_phases > 1 && ph_conc && !A->deterministic;
    if (A->col_phases > 1 && ph_conc && !A->col_concurrent) A->col_phases = 1;     // (atomic: not with spx.gpu.deterministic)
    try {
        emit_and_upload(A.get());
    } catch (const FatalError &) {
        if (A->col_phases <= 1) throw;
        A->col_phases = 1;                              // (over-long rows: no phases)
        emit_and_upload(A.get());
    }
    const bool tune_spill = spill_mode == "auto" && !A->deterministic && !A->has_symsegs;
    const bool tune_wt = wt_mode == "auto" && !A->deterministic && !A->has_symsegs;
    const bool tune_xw = xw_mode == "auto" && !A->deterministic && !sym;
    const double t_auto = now_sec();
    if (A->dev && A->nnz_stored >= 100000 && (autotune || tune_spill || tune_wt || tune_xw))
        autotune_launch(A.get(), autotune, tune_spill, tune_wt, tune_xw);
    // the read-once passes pipelined (csx_spmv_sx_kernel) against the plain read-once kernel, with whatever
    // the launch tuner settled on
    if (A->dev && sym && A->sym_pipeline == -1 && device_has_sx(A->dev)) {
        A->release_wait();
        const double t_est = device_time_spmv(A->dev, 2, 3);
        const int N = (int) std::min(100.0, std::max(8.0, 0.02 / std::max(t_est, 1e-7)));
        auto best_of = [&](bool on) {
            device_set_sx(A->dev, on);
            double best = device_time_spmv(A->dev, std::max(2, N / 10), N);
            for (int rep = 0; rep < 3; ++rep) best = std::min(best, device_time_spmv(A->dev, 0, N));
            return best;
        };
        const double t0 = best_of(false), t1 = best_of(true);
        A->sx_on = t1 < 0.985 * t0;
        device_set_sx(A->dev, A->sx_on);
        log_msg(LOG_INFO, "read-once pipeline: %s (%.2f us per SpMV with, %.2f without)\n", A->sx_on ? "on" : "off", 1e6 * t1, 1e6 * t0);
    }
    const double t_auto_end = now_sec();
    // column phases (auto): where the leftovers dominate and x is far larger than the L2 of an
    // XCD, the gathers miss it more often than not (syn-webbase: 1.6 M line fills for 2.5 M
    // gathers); slices of the columns that fit are measured against the plain stream
    if (A->dev && !sym && ph_mode == "auto" && A->n_shared == 0 && A->nnz_stored >= 100000 &&
        2 * A->n_delta_elems >= A->nnz_stored && (size_t) A->ncols * sizeof(val_t) >= ((size_t) 6 << 20)) {
        auto best_of = [&]() {
            double best = device_time_spmv(A->dev, 10, 100);
            for (int rep = 0; rep < 3; ++rep) best = std::min(best, device_time_spmv(A->dev, 0, 100));
            return best;
        };
        const double t_plain = best_of();
        // two and four slices of the columns, each on its own group of XCDs in one launch: fewer
        // line fills per gather against shorter row pieces and one more atomic hand-over of y per
        // slice (syn-webbase: 38.7 us plain, 32.6 with two, 35.6 with four, 54.6 with eight)
        size_t K = 1;
        double t_ph = t_plain;
        if (!A->deterministic) {
            for (size_t k : {(size_t) 2, (size_t) 4}) {
                if ((size_t) A->ncols * sizeof(val_t) < k * ((size_t) 3 << 19)) break;      // (slices below 1.5 MB of x: nothing to gain)
                try {
                    A->col_phases = k;
                    A->col_concurrent = true;
                    emit_and_upload(A.get());
                    const double t = best_of();
                    if (t < 0.97 * t_ph) {
                        t_ph = t;
                        K = k;
                    }
                } catch (const FatalError &) {
                    break;
                }
            }
        }
        if (A->col_phases != K || !A->col_concurrent) {
            A->col_phases = K;
            A->col_concurrent = K > 1;
            emit_and_upload(A.get());
        }
        log_msg(LOG_INFO, "column slices: %zu on XCD groups %.2f us, plain %.2f us per SpMV\n", K, 1e6 * t_ph, 1e6 * t_plain);
    }

    if (!cfg.get_bool("spx.rt.keep_encoded")) {
        std::vector<Partition> *old = new std::vector<Partition>();
        old->swap(A->parts);
        A->release_later([old] {
            parallel_for(old->size(), host_threads(), [&](size_t i) { (*old)[i] = Partition(); });
            delete old;
        });
    }
    log_msg(LOG_INFO, "launch parameters measured in %.2f s\n", t_auto_end - t_auto);
    A->emit_seconds = now_sec() - t1;
    return A.release();
}

spx_matrix_t *spx_mat_tune(spx_input_t *in, ...)
try {
    if (!in) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid input matrix");
        return SPX_INVALID_MAT;
    }
    va_list ap;
    va_start(ap, in);
    spx_option_t option = va_arg(ap, spx_option_t);
    va_end(ap);
    // RCM reordering replaces the input by P A P^T, as the reference does
    // (src/api/matvec.c:280-288: input->mat = ReorderCSR/ReorderMMF(...)); a
    // matrix that cannot be reordered is tuned in its given order.
    std::vector<idx_t> perm;
    if (option == SPX_MAT_REORDER && Config::instance().get_long("spx.rt.global_rows") > 0) {
        SETWARNING(SPX_WARN_REORDER);      // a row slice cannot be reordered on its own
        option = 0;
    }
    // spx.rt.dist_reorder: the whole matrix given to every process of a multi-GPU job -- a
    // partition-aware permutation in front of the nonzero-balanced cut (sparsex_hip.h)
    int re_mode = SPX_DIST_REORDER_RCM;
    size_t re_world = 1;
    {
        const std::string dr = Config::instance().get_str("spx.rt.dist_reorder");
        const long world = Config::instance().get_long("spx.rt.gpu_world");
        if (dr != "none" && dr != "rcm" && dr != "rcm_owner") {
            SETERROR_1(SPX_ERR_ARG_INVALID, "spx.rt.dist_reorder: none, rcm or rcm_owner");
            return SPX_INVALID_MAT;
        }
        if (dr != "none" && world > 1 && Config::instance().get_long("spx.rt.global_rows") <= 0) {
            option = SPX_MAT_REORDER;
            re_mode = dr == "rcm_owner" ? SPX_DIST_REORDER_RCM_OWNER : SPX_DIST_REORDER_RCM;
            re_world = (size_t) world;
        }
    }
    if (option == SPX_MAT_REORDER) {
        MatrixInput *re = nullptr;
        try {
            re = reorder_rcm(*in->mat, perm, re_mode, re_world);
        } catch (const FatalError &) {
            re = nullptr;
            perm.clear();
        }
        if (re) {
            delete in->mat;
            in->mat = re;
        } else {
            SETWARNING(SPX_WARN_REORDER);
        }
    }
    spx_matrix_t *A = SPX_INVALID_MAT;
    try {
        A = do_tune(in);
    } catch (const FatalError &e) {
        SETERROR_0(SPX_ERR_TUNED_MAT);
        return SPX_INVALID_MAT;
    }
    if (!perm.empty()) {
        A->permutation = (spx_perm_t *) malloc(perm.size() * sizeof(spx_perm_t));
        std::copy(perm.begin(), perm.end(), A->permutation);
    }
    return A;
} SPX_C_BOUNDARY(return SPX_INVALID_MAT;)

spx_error_t spx_mat_destroy(spx_matrix_t *A)
try {
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle");
        return SPX_FAILURE;
    }
    try { device_free(A->dev); } catch (...) {}
    try { dist_free_plan(A->dist); } catch (...) {}
    if (A->permutation) free(A->permutation);
    delete A;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

// ---- get / set entry ------------------------------------------------------------------
// Random access into the tuned matrix (reference: src/api/matvec.c:324-407,
// include/sparsex/internals/CsxGetSet.hpp:195-320).  The reference walks the
// ctl stream of the row and of the rows above it within `span`.  Here the
// row-block that owns the row is decoded on the host from the stream's index
// arrays (stream_index.cpp) and the value is read or written where it lives:
// in HBM, or in the host copy of a host-only matrix.  A restored matrix works
// like a freshly tuned one.  Where the encoded partitions are still held
// (spx.rt.keep_encoded, the default) they are kept in step, so that an export
// in the reference's CSX layout shows the change.

}  // extern "C"

namespace {

// rows a unit anchored in (row) reaches below its anchor
idx_t unit_span(const Elem &e)
{
    if (!e.is_unit()) return 0;
    if (e.type == ENC_V || e.type == ENC_D || e.type == ENC_AD)
        return (idx_t)(e.size - 1) * (idx_t) e.delta;
    if (enc_is_block_row(e.type)) return (idx_t) enc_block_align(e.type) - 1;
    if (enc_is_block_col(e.type)) return (idx_t) e.size / enc_block_align(e.type) - 1;
    return 0;
}

void build_spans(spx_matrix_t *A)
{
    if (A->spans.size() == A->parts.size()) return;
    A->spans.assign(A->parts.size(), std::vector<idx_t>());
    A->max_span.assign(A->parts.size(), 0);
    for (size_t p = 0; p < A->parts.size(); ++p) {
        const Partition &pt = A->parts[p];
        std::vector<idx_t> &sp = A->spans[p];
        sp.assign(pt.rowptr.size() - 1, 0);
        for (size_t i = 0; i + 1 < pt.rowptr.size(); ++i)
            for (idx_t j = pt.rowptr[i]; j < pt.rowptr[i + 1]; ++j)
                sp[i] = std::max(sp[i], unit_span(pt.elems[j]));
        for (idx_t v : sp) A->max_span[p] = std::max(A->max_span[p], v);
    }
}

// pointer to the stored value of (row, col) in the encoded partitions,
// 1-based global; NULL if absent
val_t *locate_encoded(spx_matrix_t *A, idx_t row, idx_t col)
{
    if (A->symmetric) {
        if (col > row) std::swap(row, col);
        if (row == col) {
            for (size_t p = 0; p < A->parts.size(); ++p) {
                const PartBounds &b = A->bounds[A->first_part + p];
                idx_t r = row - 1 - b.row_start;
                if (r >= 0 && r < b.nr_rows && (size_t) r < A->diag[p].size()) return &A->diag[p][r];
            }
            return nullptr;
        }
    }
    for (size_t p = 0; p < A->parts.size(); ++p) {
        const PartBounds &b = A->bounds[A->first_part + p];
        idx_t r = row - b.row_start;          // 1-based inside the partition
        if (r < 1 || r > b.nr_rows) continue;
        Partition &pt = A->parts[p];
        const std::vector<idx_t> &sp = A->spans[p];
        const idx_t nr = (idx_t) pt.rowptr.size() - 1;
        for (idx_t i = std::min(r, nr); i >= 1 && r - i <= A->max_span[p]; --i) {
            if (sp[(size_t) i - 1] < r - i) continue;   // nothing anchored in row i reaches row r
            for (idx_t j = pt.rowptr[i - 1]; j < pt.rowptr[i]; ++j) {
                Elem &e = pt.elems[j];
                if (!e.is_unit()) {
                    if (e.row == r && e.col == col) return &e.val;
                    continue;
                }
                for (size_t k = 0; k < e.size; ++k) {
                    idx_t er, ec;
                    unit_elem_coords(e, k, er, ec);
                    if (er == r && ec == col) return &pt.pool[e.voff + k];
                }
            }
        }
        return nullptr;
    }
    return nullptr;
}

// Where the descriptor stream holds (row, col), 1-based global: positions in
// its values (a symmetric matrix holds an off-diagonal entry once in a tile, or
// twice -- lower triangle and mirror image), or the diagonal entry.
struct EntryRef {
    bool diagonal = false;
    size_t diag_row = 0;
    std::vector<size_t> pos;
    std::vector<size_t> mirror_pos;     // symmetric slice: copies in the thin mirror list
    bool found() const { return diagonal || !pos.empty(); }
};

EntryRef locate_stream(const spx_matrix_t *A, idx_t row, idx_t col)
{
    EntryRef ref;
    const GpuStream *s = A->host_stream ? A->host_stream.get() : A->index.get();
    if (!s) return ref;
    if (A->symmetric && row == col) {
        if (row - 1 >= A->own_lo && row - 1 < A->own_hi) {
            ref.diagonal = true;
            ref.diag_row = (size_t) row - 1;
        }
        return ref;
    }
    stream_locate(*s, row - 1, col - 1, ref.pos);
    if (A->symmetric) {
        stream_locate(*s, col - 1, row - 1, ref.pos);
        // (the stored entry is the lower one; its mirror image may sit in the thin list)
        stream_locate_mirror(*s, std::min(row, col) - 1, std::max(row, col) - 1, ref.mirror_pos);
    }
    return ref;
}

bool entry_args(const spx_matrix_t *A, spx_option_t indexing, spx_index_t &row, spx_index_t &col)
{
    const int base = (indexing == SPX_INDEX_ONE_BASED) ? 1 : 0;
    if (row - base < 0 || row - base >= A->nrows || col - base < 0 || col - base >= A->ncols) {
        SETERROR_0(SPX_OUT_OF_BOUNDS);
        return false;
    }
    row = row - base;
    col = col - base;
    if (A->permutation != SPX_INVALID_PERM) {   // reordered matrix: src/api/matvec.c:351-354
        row = A->permutation[row];
        col = A->permutation[col];
    }
    row += 1;                  // internally 1-based
    col += 1;
    return true;
}

}  // namespace

extern "C" {

spx_error_t spx_mat_get_entry(const spx_matrix_t *A_, spx_index_t row, spx_index_t column,
                              spx_value_t *value, ...)
try {
    va_list ap;
    va_start(ap, value);
    spx_option_t indexing = va_arg(ap, spx_option_t);
    va_end(ap);
    spx_matrix_t *A = const_cast<spx_matrix_t *>(A_);
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle");
        return SPX_FAILURE;
    }
    if (!value) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid value pointer");
        return SPX_FAILURE;
    }
    if (!entry_args(A, indexing, row, column)) return SPX_FAILURE;
    std::lock_guard<std::mutex> lk(A->mtx);
    const EntryRef ref = locate_stream(A, row, column);
    if (!ref.found()) {
        SETERROR_0(SPX_ERR_ENTRY_NOT_FOUND);
        return SPX_FAILURE;
    }
    try {
        if (A->host_stream)
            *value = ref.diagonal ? A->host_stream->dvalues[ref.diag_row] : A->host_stream->values[ref.pos[0]];
        else
            *value = ref.diagonal ? device_peek(A->dev, true, ref.diag_row)
                                  : device_peek(A->dev, false, ref.pos[0]);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_mat_set_entry(spx_matrix_t *A, spx_index_t row, spx_index_t column,
                              spx_value_t value, ...)
try {
    va_list ap;
    va_start(ap, value);
    spx_option_t indexing = va_arg(ap, spx_option_t);
    va_end(ap);
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle");
        return SPX_FAILURE;
    }
    if (!entry_args(A, indexing, row, column)) {
        SETWARNING(SPX_WARN_ENTRY_NOT_SET);
        return SPX_FAILURE;
    }
    std::lock_guard<std::mutex> lk(A->mtx);
    const EntryRef ref = locate_stream(A, row, column);
    if (!ref.found()) {
        SETERROR_0(SPX_ERR_ENTRY_NOT_FOUND);
        return SPX_FAILURE;
    }
    try {
        if (A->host_stream) {
            if (ref.diagonal) A->host_stream->dvalues[ref.diag_row] = value;
            for (size_t p : ref.pos) A->host_stream->values[p] = value;
            for (size_t p : ref.mirror_pos) A->host_stream->mirror_val[p] = value;
        } else {
            if (ref.diagonal) device_poke(A->dev, true, ref.diag_row, value);
            for (size_t p : ref.pos) device_poke(A->dev, false, p, value);
            for (size_t p : ref.mirror_pos) device_poke_mirror(A->dev, p, value);
        }
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    if (!A->parts.empty()) {
        // the encoded partitions follow (export in the reference's layout)
        build_spans(A);
        val_t *v = locate_encoded(A, row, column);
        if (v) *v = value;
        A->exported.clear();
        A->exported_rows_info.clear();
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

// ---- save / restore -----------------------------------------------------------------
// The reference archives every partition's ctl/values/id_map/rows_info with
// Boost.Serialization and re-runs its JIT on restore
// (include/sparsex/internals/CsxSaveRestore.hpp:77-371).  Here the file holds the
// row-block descriptor stream as it lies in HBM plus the matrix header, so a
// restore is a read and an upload -- no preprocessing.

}  // extern "C"

namespace {

const char kMagic[8] = {'S', 'P', 'X', 'H', 'I', 'P', '1', '4'};

template <typename T, typename A>
bool put_vec(FILE *f, const std::vector<T, A> &v)
{
    uint64_t n = v.size();
    if (fwrite(&n, sizeof(n), 1, f) != 1) return false;
    return n == 0 || fwrite(v.data(), sizeof(T), n, f) == n;
}

template <typename T, typename A>
bool get_vec(FILE *f, std::vector<T, A> &v)
{
    uint64_t n = 0;
    if (fread(&n, sizeof(n), 1, f) != 1) return false;
    if (n > (uint64_t) 1 << 40) return false;
    // (a count that the rest of the file cannot hold is a damaged file, not a
    // reason to allocate terabytes)
    const long here = ftell(f);
    if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
        const long end = ftell(f);
        if (fseek(f, here, SEEK_SET) != 0 || end < here ||
            n > (uint64_t)(end - here) / sizeof(T))
            return false;
    }
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

struct SavedHeader {
    int32_t nrows, ncols, nnz, symmetric;
    uint64_t nr_partitions, first_part, last_part;
    int32_t own_lo, own_hi;
    uint64_t nnz_stored, n_unit_elems, n_delta_elems, n_units;
    uint32_t n_carry, pad;
    uint32_t n_spill, lds_doubles;
    uint32_t waves, n_encoded;          // n_encoded: encoded partitions that follow the stream
    uint32_t sym_atomic, pad3;          // symmetric tiles: sums handed over with global atomics
    uint64_t checksum;                  // FNV-1a over the index arrays
};

// a cheap guard against damaged or stale files: the kernels trust the index
// arrays, so they are summed (the values are not: a flipped value bit gives a
// wrong product, not an out-of-bounds access)
template <typename T>
void fnv(uint64_t &h, const std::vector<T> &v)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(v.data());
    for (size_t i = 0, n = v.size() * sizeof(T); i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
}

uint64_t stream_checksum(const GpuStream &s)
{
    uint64_t h = 0xCBF29CE484222325ull;
    fnv(h, s.rbs); fnv(h, s.passes); fnv(h, s.descs); fnv(h, s.cidx); fnv(h, s.segrows);
    fnv(h, s.shared); fnv(h, s.fix_ptr); fnv(h, s.fix_idx); fnv(h, s.slot_group_col);
    fnv(h, s.mirror_rows); fnv(h, s.mirror_ptr); fnv(h, s.mirror_col);
    return h;
}

struct SavedPartition {
    uint64_t nr_rows, nr_cols, nnz, elems_size;
    int32_t type, row_start;
};

bool put_partition(FILE *f, const Partition &p, const std::vector<val_t> *diag)
{
    SavedPartition h;
    memset(&h, 0, sizeof(h));
    h.nr_rows = p.nr_rows; h.nr_cols = p.nr_cols; h.nnz = p.nnz; h.elems_size = p.elems_size;
    h.type = p.type; h.row_start = p.row_start;
    std::vector<Elem> live(p.elems.begin(), p.elems.begin() + p.elems_size);
    static const std::vector<val_t> none;
    return fwrite(&h, sizeof(h), 1, f) == 1 && put_vec(f, live) && put_vec(f, p.rowptr) &&
           put_vec(f, p.pool) && put_vec(f, diag ? *diag : none);
}

bool get_partition(FILE *f, Partition &p, std::vector<val_t> &diag)
{
    SavedPartition h;
    if (fread(&h, sizeof(h), 1, f) != 1) return false;
    p.nr_rows = h.nr_rows; p.nr_cols = h.nr_cols; p.nnz = h.nnz; p.elems_size = h.elems_size;
    p.type = h.type; p.row_start = h.row_start;
    if (!(get_vec(f, p.elems) && get_vec(f, p.rowptr) && get_vec(f, p.pool) && get_vec(f, diag)))
        return false;
    if (p.elems.size() != p.elems_size || p.rowptr.empty() || p.rowptr.size() > p.nr_rows + 1)
        return false;
    for (size_t i = 0; i + 1 < p.rowptr.size(); ++i)
        if (p.rowptr[i] > p.rowptr[i + 1]) return false;
    if ((size_t) p.rowptr.back() != p.elems_size || p.rowptr[0] != 0) return false;
    for (const Elem &e : p.elems) {
        if (e.size == 0 || e.row < 1 || (size_t) e.row > p.nr_rows || e.col < 1 || (size_t) e.col > p.nr_cols)
            return false;
        if (e.is_unit() && ((size_t) e.voff + e.size > p.pool.size() || e.type <= ENC_NONE || e.type >= ENC_MAX))
            return false;
    }
    return true;
}

}  // namespace

extern "C" {


spx_error_t spx_mat_save(const spx_matrix_t *A, const char *filename)
try {
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle");
        return SPX_FAILURE;
    }
    if (!filename) {
        SETWARNING(SPX_WARN_CSXFILE);
        filename = "csx_file";
    }
    // the file must show what the matrix holds now (the reference writes
    // set_entry straight into the arrays it archives, src/api/matvec.c:409-425)
    GpuStream tmp;
    const GpuStream *gs = A->host_stream.get();
    if (!gs) {
        if (!A->dev) {
            SETERROR_1(SPX_ERR_TUNED_MAT, "matrix holds no descriptor stream");
            return SPX_FAILURE;
        }
        try {
            device_download(A->dev, tmp);
        } catch (const FatalError &e) {
            SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
            return SPX_FAILURE;
        }
        gs = &tmp;
    }
    // (the SPX_ERR_FILE_* codes are "system" errors on which the default
    // handler exits, as in the reference; a bad file name is not worth that)
    FILE *f = fopen(filename, "wb");
    if (!f) {
        SETERROR_1(SPX_ERR_FILE, "cannot open file for writing");
        return SPX_FAILURE;
    }
    SavedHeader h;
    memset(&h, 0, sizeof(h));
    h.nrows = A->nrows; h.ncols = A->ncols; h.nnz = A->nnz; h.symmetric = A->symmetric;
    h.nr_partitions = A->nr_partitions; h.first_part = A->first_part; h.last_part = A->last_part;
    h.own_lo = A->own_lo; h.own_hi = A->own_hi;
    h.nnz_stored = A->nnz_stored; h.n_unit_elems = A->n_unit_elems;
    h.n_delta_elems = A->n_delta_elems; h.n_units = A->n_units;
    h.n_carry = gs->n_carry;
    h.pad = (gs->sym_fused ? 1u : 0u) | (gs->pass_stride << 1);
    h.n_spill = gs->n_spill;
    h.lds_doubles = gs->lds_doubles;
    h.waves = gs->waves;
    h.n_encoded = (uint32_t) A->parts.size();
    h.sym_atomic = gs->sym_atomic ? 1u : 0u;
    // (bit 2: the product runs with the unit windows of x in LDS; bits 8-15 / 16-31: their gap and budget)
    h.pad3 = (gs->deterministic ? 1u : 0u) | (gs->wave_tiles ? 2u : 0u);
    if (gs->sx_on) h.pad3 |= 8u;                       // (bit 3: the read-once passes run pipelined)
    if (gs->xw_on) h.pad3 |= 4u | ((gs->xw_gap & 255u) << 8) | (std::min<uint32_t>(gs->xw_budget, 65535u) << 16);
    h.checksum = stream_checksum(*gs);
    bool good = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&h, sizeof(h), 1, f) == 1;
    std::vector<int32_t> bnd;
    for (const PartBounds &b : A->bounds) {
        bnd.push_back(b.row_start);
        bnd.push_back(b.nr_rows);
        bnd.push_back((int32_t) b.nnz);
    }
    good = good && put_vec(f, bnd) && put_vec(f, gs->rbs) && put_vec(f, gs->passes) &&
           put_vec(f, gs->descs) && put_vec(f, gs->cidx) &&
           put_vec(f, gs->segrows) && put_vec(f, gs->shared) && put_vec(f, gs->dvalues) &&
           put_vec(f, gs->values) && put_vec(f, gs->fix_ptr) && put_vec(f, gs->fix_idx) &&
           put_vec(f, gs->slot_group_col) && put_vec(f, gs->mirror_rows) && put_vec(f, gs->mirror_ptr) &&
           put_vec(f, gs->mirror_col) && put_vec(f, gs->mirror_val);
    std::vector<int32_t> perm;
    if (A->permutation) perm.assign(A->permutation, A->permutation + A->nrows);
    good = good && put_vec(f, perm);
    // the encoded partitions, where they are still held: a restored matrix can
    // then be exported in the reference's CSX layout like the one that was saved
    for (size_t i = 0; good && i < A->parts.size(); ++i)
        good = put_partition(f, A->parts[i], A->symmetric ? &A->diag[i] : nullptr);
    good = (fclose(f) == 0) && good;
    if (!good) {
        SETERROR_1(SPX_ERR_FILE, "writing the tuned matrix failed");
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_matrix_t *spx_mat_restore(const char *filename)
try {
    if (!filename) {
        SETERROR_0(SPX_ERR_FILE);
        return SPX_INVALID_MAT;
    }
    if (access(filename, F_OK | R_OK) == -1) {
        SETERROR_0(SPX_ERR_FILE);
        return SPX_INVALID_MAT;
    }
    FILE *f = fopen(filename, "rb");
    if (!f) {
        SETERROR_0(SPX_ERR_FILE);
        return SPX_INVALID_MAT;
    }
    char magic[8];
    SavedHeader h;
    std::unique_ptr<GpuStream> gs(new GpuStream);
    std::vector<int32_t> bnd;
    bool good = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0 &&
                fread(&h, sizeof(h), 1, f) == 1 && get_vec(f, bnd) && get_vec(f, gs->rbs) &&
                get_vec(f, gs->passes) && get_vec(f, gs->descs) &&
                get_vec(f, gs->cidx) && get_vec(f, gs->segrows) && get_vec(f, gs->shared) &&
                get_vec(f, gs->dvalues) && get_vec(f, gs->values) && get_vec(f, gs->fix_ptr) &&
                get_vec(f, gs->fix_idx) && get_vec(f, gs->slot_group_col) && get_vec(f, gs->mirror_rows) &&
                get_vec(f, gs->mirror_ptr) && get_vec(f, gs->mirror_col) && get_vec(f, gs->mirror_val);
    std::vector<int32_t> perm;
    good = good && get_vec(f, perm);
    std::unique_ptr<matrix> A(new matrix);
    if (good && h.n_encoded) {
        good = h.n_encoded == h.last_part - h.first_part && h.n_encoded <= (1u << 20);
        if (good) {
            A->parts.resize(h.n_encoded);
            if (h.symmetric) A->diag.resize(h.n_encoded);
        }
        std::vector<val_t> nodiag;
        for (uint32_t i = 0; good && i < h.n_encoded; ++i)
            good = get_partition(f, A->parts[i], h.symmetric ? A->diag[i] : nodiag);
    }
    fclose(f);
    std::string why = "not a tuned-matrix file of this build";
    if (good) {
        good = h.nrows >= 0 && h.ncols >= 0 && bnd.size() == 3 * h.nr_partitions &&
               h.first_part <= h.last_part && h.last_part <= h.nr_partitions &&
               (perm.empty() || perm.size() == (size_t) h.nrows) && h.own_lo >= 0 &&
               h.own_lo <= h.own_hi && h.own_hi <= h.nrows &&
               (!h.symmetric || gs->dvalues.size() == (size_t) h.nrows);
        for (int32_t v : perm) good = good && v >= 0 && v < h.nrows;
    }
    if (good) {
        gs->n_carry = h.n_carry;
        gs->sym_fused = (h.pad & 1u) != 0;
        gs->pass_stride = h.pad >> 1;
        gs->n_spill = h.n_spill;
        gs->lds_doubles = h.lds_doubles;
        gs->waves = h.waves;
        gs->sym_atomic = h.sym_atomic != 0;
        gs->deterministic = (h.pad3 & 1u) != 0;
        gs->wave_tiles = (h.pad3 & 2u) != 0;
        gs->xw_on = (h.pad3 & 4u) != 0;
        gs->sx_on = (h.pad3 & 8u) != 0;
        gs->sx_plan = gs->sx_on;
        gs->xw_gap = gs->xw_on ? ((h.pad3 >> 8) & 255u) : 16u;
        gs->xw_budget = gs->xw_on ? (h.pad3 >> 16) : 0u;
        gs->nnz_stored = h.nnz_stored; gs->n_unit_elems = h.n_unit_elems;
        gs->n_delta_elems = h.n_delta_elems; gs->n_units = h.n_units;
        good = stream_checksum(*gs) == h.checksum;
        if (!good) why = "tuned-matrix file is damaged (checksum of the index arrays)";
        else good = stream_validate(*gs, (size_t) h.nrows, (size_t) h.ncols, gs->values.size(), why);
    }
    if (!good) {
        SETERROR_1(SPX_ERR_FILE, why.c_str());
        return SPX_INVALID_MAT;
    }
    A->nrows = h.nrows; A->ncols = h.ncols; A->nnz = h.nnz; A->symmetric = h.symmetric;
    A->permutation = SPX_INVALID_PERM;
    if (!perm.empty()) {
        A->permutation = (spx_perm_t *) malloc(perm.size() * sizeof(spx_perm_t));
        std::copy(perm.begin(), perm.end(), A->permutation);
    }
    A->nr_partitions = h.nr_partitions; A->first_part = h.first_part; A->last_part = h.last_part;
    for (size_t i = 0; i < h.nr_partitions; ++i)
        A->bounds.push_back(PartBounds{bnd[3 * i], bnd[3 * i + 1], (size_t) bnd[3 * i + 2]});
    A->full_colind = false;
    A->dev = nullptr;
    A->waves = (h.waves == 2 || h.waves == 8) ? (int) h.waves : 4;
    A->own_lo = h.own_lo; A->own_hi = h.own_hi;
    A->nnz_stored = h.nnz_stored; A->n_unit_elems = h.n_unit_elems;
    A->n_delta_elems = h.n_delta_elems; A->n_units = h.n_units;
    A->value_bytes = gs->values.size() * sizeof(val_t);
    A->index_bytes = gs->index_bytes();
    A->n_rowblocks = gs->rbs.size();
    A->n_shared = gs->shared.size();
    A->has_tiles = stream_has_tiles(*gs);
    A->has_symtiles = stream_has_pass(*gs, SPX_PASS_SYMTILE);
    A->has_symsegs = stream_has_symsegs(*gs);
    A->sym_atomic = gs->sym_atomic;
    A->deterministic = gs->deterministic;
    A->wave_tiles = gs->wave_tiles ? 1 : 0;
    A->xw_on = gs->xw_on;
    A->unit_windows = gs->xw_on ? 1 : 0;
    A->sx_on = gs->sx_on;
    A->sym_pipeline = gs->sx_on ? 1 : 0;
    A->xw_budget = gs->xw_budget;
    A->xw_gap = gs->xw_gap;
    {
        // (column slices: what spx_hip_mat_info reports comes from the flags the stream carries)
        size_t slices = 1;
        for (size_t i = 1; i < gs->rbs.size(); ++i) slices += (gs->rbs[i].flags & SPX_RB_PHASE_START) ? 1 : 0;
        A->col_phases = slices;
        A->col_concurrent = slices > 1 && !gs->rbs.empty() && (gs->rbs[0].flags & SPX_RB_ACCUM);
    }
    A->tune_seconds = 0.0;
    A->auto_rb = false;
    const double t0 = now_sec();
    Config &cfg = Config::instance();
    A->host_only = cfg.get_bool("spx.rt.host_only");
    A->device_ordinal = (int) cfg.get_long("spx.rt.device");
    A->full_colind = cfg.get_bool("spx.matrix.full_colind");
    if (A->symmetric && !gs->sym_fused) stream_touched_rows(*gs, A->own_lo, A->conflict_rows);
    if (A->own_lo > 0 || A->own_hi < A->nrows) stream_read_cols(*gs, A->own_lo, A->own_hi, (size_t) A->ncols, A->halo_cols);
    A->first_block_row = A->own_lo;
    for (const SpxRowBlock &rb : gs->rbs) A->first_block_row = std::min<idx_t>(A->first_block_row, (idx_t) rb.row0);
    try {
        if (!A->host_only) {
            A->dev = device_upload(*gs, (size_t) A->nrows, (size_t) A->ncols, A->symmetric != 0,
                                   A->own_lo, A->own_hi, A->device_ordinal);
            keep_index(A.get(), std::move(*gs));
        } else {
            A->host_stream = std::move(gs);
        }
    } catch (const FatalError &) {
        SETERROR_0(SPX_ERR_TUNED_MAT);
        return SPX_INVALID_MAT;
    }
    A->emit_seconds = now_sec() - t0;
    return A.release();
} SPX_C_BOUNDARY(return SPX_INVALID_MAT;)

spx_index_t spx_mat_get_nrows(const spx_matrix_t *A)
try {
    if (!A) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle"); return SPX_FAILURE; }
    return A->nrows;
} SPX_C_BOUNDARY(return 0;)

spx_index_t spx_mat_get_ncols(const spx_matrix_t *A)
try {
    if (!A) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle"); return SPX_FAILURE; }
    return A->ncols;
} SPX_C_BOUNDARY(return 0;)

spx_index_t spx_mat_get_nnz(const spx_matrix_t *A)
try {
    if (!A) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle"); return SPX_FAILURE; }
    return A->nnz;
} SPX_C_BOUNDARY(return 0;)

static spx_partition_t *part_alloc(size_t n)
{
    spx_partition_t *p = (spx_partition_t *) malloc(sizeof(spx_partition_t));
    p->nr_partitions = n;
    p->parts = NULL;
    p->nodes = NULL;
    p->affinity = NULL;
    p->row_start = (spx_index_t *) malloc(sizeof(spx_index_t) * (n ? n : 1));
    p->row_end = (spx_index_t *) malloc(sizeof(spx_index_t) * (n ? n : 1));
    return p;
}

spx_partition_t *spx_mat_get_partition(const spx_matrix_t *A)
try {
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle");
        return SPX_INVALID_PART;
    }
    spx_partition_t *p = part_alloc(A->nr_partitions);
    for (size_t i = 0; i < A->nr_partitions; ++i) {
        p->row_start[i] = A->bounds[i].row_start;
        p->row_end[i] = A->bounds[i].row_start + A->bounds[i].nr_rows;
    }
    return p;
} SPX_C_BOUNDARY(return SPX_INVALID_PART;)

spx_index_t *spx_partition_get_rs(const spx_partition_t *p)
try {
    if (!p) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid partition handle"); return NULL; }
    return p->row_start;
} SPX_C_BOUNDARY(return nullptr;)

spx_index_t *spx_partition_get_re(const spx_partition_t *p)
try {
    if (!p) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid partition handle"); return NULL; }
    return p->row_end;
} SPX_C_BOUNDARY(return nullptr;)

spx_perm_t *spx_mat_get_perm(const spx_matrix_t *A)
try {
    if (!A) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle");
        return SPX_INVALID_PERM;
    }
    if (!A->permutation) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "a permutation is not available");
        return SPX_INVALID_PERM;
    }
    return A->permutation;
} SPX_C_BOUNDARY(return SPX_INVALID_PERM;)

spx_partition_t *spx_partition_csr(const spx_index_t *rowptr, spx_index_t nr_rows,
                                   size_t nr_threads)
try {
    // The reference's split of a CSR matrix over threads (src/api/matvec.c:689-737), as cuts of the row
    // pointer: every split takes rows until it holds a quota of (rowptr[n] - 1) / T nonzeros; what is left
    // behind the last full split is one more, open, split.  Observable quirks kept: the open split ends at
    // row n + 1, not n (clients only ever use it as an upper bound of a loop that also stops at n), and when
    // the quota is 0 every row is a split of its own and there is no open one.
    spx_partition_t *ret = part_alloc(nr_threads);
    const spx_index_t *const ends = rowptr + 1;                 // ends[r]: nonzeros up to and including row r
    const size_t quota = (size_t) (rowptr[nr_rows] - 1) / nr_threads;
    size_t k = 0;
    spx_index_t begin = 0;
    ret->row_start[0] = 0;
    while (k < nr_threads && begin < nr_rows) {
        // the first row whose end brings the split up to its quota
        const spx_index_t *hit = std::lower_bound(ends + begin, ends + nr_rows, (spx_index_t) (rowptr[begin] + (spx_index_t) quota));
        if (hit == ends + nr_rows) break;
        begin = (spx_index_t) (hit - ends) + 1;
        ret->row_end[k++] = begin;
        if (k < nr_threads) ret->row_start[k] = begin;
    }
    if (k < nr_threads && (size_t) (rowptr[nr_rows] - rowptr[begin]) < quota) ret->row_end[k] = nr_rows + 1;
    return ret;
} SPX_C_BOUNDARY(return SPX_INVALID_PART;)

spx_error_t spx_partition_destroy(spx_partition_t *p)
try {
    if (!p) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid partition handle");
        return SPX_FAILURE;
    }
    free(p->parts); free(p->nodes); free(p->affinity);
    free(p->row_start); free(p->row_end);
    free(p);
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

// ======================================================================================
//  options
// ======================================================================================

void spx_option_set(const char *option, const char *value)
try {
    if (!option || !value) { SETWARNING(SPX_WARN_TUNING_OPT); return; }
    try {
        Config::instance().set(option, value);
    } catch (const FatalError &) {
        exit(1);    // invalid enumerated value: the reference exits (Encodings.cpp:171-186)
    }
} SPX_C_BOUNDARY(return;)

void spx_options_set_from_env()
try {
    try {
        Config::instance().load_from_env();
    } catch (const FatalError &) {
        exit(1);
    }
} SPX_C_BOUNDARY(return;)

void spx_hip_options_reset(void) { Config::instance().reset_defaults(); }

spx_error_t spx_hip_dist_reorder(const spx_index_t *rowptr, const spx_index_t *colind, spx_index_t nrows,
                                 int indexing, int world, int mode, int flags, spx_index_t *perm)
try {
    if (!rowptr || !colind || !perm || nrows < 0 || world < 1 ||
        (indexing != SPX_INDEX_ZERO_BASED && indexing != SPX_INDEX_ONE_BASED) ||
        (mode != SPX_DIST_REORDER_RCM && mode != SPX_DIST_REORDER_RCM_OWNER)) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument");
        return SPX_FAILURE;
    }
    try {
        std::vector<idx_t> p;
        dist_reorder_csr(rowptr, colind, (size_t) nrows, indexing == SPX_INDEX_ZERO_BASED,
                         (flags & SPX_DIST_PATTERN_SYMMETRIC) != 0, (size_t) world, mode, p);
        std::copy(p.begin(), p.end(), perm);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_ARG_INVALID, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

void spx_hip_xform(int from, int to, spx_index_t *row, spx_index_t *col,
                   spx_index_t nr_rows, spx_index_t nr_cols)
try {
    xform(from, to, *row, *col, nr_rows, nr_cols);
} SPX_C_BOUNDARY(return;)

// ======================================================================================
//  SpMV
// ======================================================================================

static spx_error_t check_mv(const spx_matrix_t *A, const spx_vector_t *x, spx_vector_t *y)
{
    if (!A) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle"); return SPX_FAILURE; }
    if (!x) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector x"); return SPX_FAILURE; }
    if (!y) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector y"); return SPX_FAILURE; }
    // Either vector of the wrong length is rejected.  (The reference tests
    // `!x_ok && !y_ok`, src/api/matvec.c:571, and so lets a single mismatch
    // through to an out-of-bounds access; a GPU kernel must not do that.)
    if (!check_vec_dim(x, (unsigned long) A->ncols) ||
        !check_vec_dim(y, (unsigned long) A->nrows)) {
        SETERROR_0(SPX_ERR_DIM);
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
}

static spx_error_t run_host(const spx_matrix_t *A, spx_value_t alpha, const spx_vector_t *x,
                            spx_value_t beta, spx_vector_t *y)
{
    if (!A->dev) {
        SETERROR_1(SPX_ERR_TUNED_MAT,
                   "matrix was tuned with spx.rt.host_only=true: no HIP executor");
        return SPX_FAILURE;
    }
    try {
        // a row-partitioned matrix with an exchange plan: every process ends up
        // with all of y in host memory, as the reference's caller expects
        std::function<void(double *, void *)> after;
        if (A->dist) after = [A](double *d_y, void *st) { dist_complete(A->dist, d_y, true, false, st); };
        const bool x_locked = vec_page_locked(x), y_locked = vec_page_locked(y);
        device_set_host_parts(A->dev, (size_t) std::max<long>(0, Config::instance().get_long("spx.rt.host_parts")));
        device_spmv_host(A->dev, alpha, x->elements, x_locked, beta, y->elements, y_locked, after, vec_version(x));
        vec_touch(y);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
}

spx_error_t spx_matvec_mult(spx_value_t alpha, const spx_matrix_t *A, const spx_vector_t *x,
                            spx_vector_t *y)
try {
    if (check_mv(A, x, y) != SPX_SUCCESS) return SPX_FAILURE;
    return run_host(A, alpha, x, 0.0, y);
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_matvec_kernel(spx_value_t alpha, const spx_matrix_t *A, const spx_vector_t *x,
                              spx_value_t beta, spx_vector_t *y)
try {
    if (check_mv(A, x, y) != SPX_SUCCESS) return SPX_FAILURE;
    return run_host(A, alpha, x, beta, y);
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_matvec_kernel_csr(spx_matrix_t **A, spx_index_t nrows, spx_index_t ncols,
                                  const spx_index_t *rowptr, const spx_index_t *colind,
                                  const spx_value_t *values, spx_value_t alpha,
                                  const spx_vector_t *x, spx_value_t beta, spx_vector_t *y)
try {
    if (!x) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector x"); return SPX_FAILURE; }
    if (!y) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector y"); return SPX_FAILURE; }
    if (!A) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle"); return SPX_FAILURE; }
    if (!*A) {
        if (!check_mat_dim(nrows) || !check_mat_dim(ncols)) {
            SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix dimensions");
            return SPX_FAILURE;
        }
        if (!rowptr) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid rowptr argument"); return SPX_FAILURE; }
        if (!colind) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid colind argument"); return SPX_FAILURE; }
        if (!values) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid values argument"); return SPX_FAILURE; }
        spx_input_t *in = spx_input_load_csr(rowptr, colind, values, nrows, ncols,
                                             SPX_INDEX_ZERO_BASED);
        if (!in) return SPX_FAILURE;
        *A = spx_mat_tune(in, 0);
        spx_input_destroy(in);
        if (!*A) return SPX_FAILURE;
    }
    return spx_matvec_kernel(alpha, *A, x, beta, y);
} SPX_C_BOUNDARY(return SPX_FAILURE;)

static spx_error_t check_dev(const spx_matrix_t *A, const void *x, const void *y)
{
    if (!A) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid matrix handle"); return SPX_FAILURE; }
    if (!x) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector x"); return SPX_FAILURE; }
    if (!y) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector y"); return SPX_FAILURE; }
    if (!A->dev) {
        SETERROR_1(SPX_ERR_TUNED_MAT,
                   "matrix was tuned with spx.rt.host_only=true: no HIP executor");
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
}

spx_error_t spx_hip_matvec_mult(spx_value_t alpha, const spx_matrix_t *A,
                                const spx_value_t *x_dev, spx_value_t *y_dev, void *stream)
try {
    return spx_hip_matvec_kernel(alpha, A, x_dev, 0.0, y_dev, stream);
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_matvec_kernel(spx_value_t alpha, const spx_matrix_t *A,
                                  const spx_value_t *x_dev, spx_value_t beta,
                                  spx_value_t *y_dev, void *stream)
try {
    if (check_dev(A, x_dev, y_dev) != SPX_SUCCESS) return SPX_FAILURE;
    try {
        device_spmv(A->dev, alpha, x_dev, beta, y_dev, stream);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_matvec_parts(spx_value_t alpha, const spx_matrix_t *A, const spx_value_t *x_dev,
                                 spx_value_t beta, spx_value_t *y_dev, int parts, void *stream, int *launched)
try {
    if (check_dev(A, x_dev, y_dev) != SPX_SUCCESS) return SPX_FAILURE;
    try {
        std::vector<size_t> bounds;
        // (a cut of its own, slot 1: the one an attached exchange plan walks is left alone)
        const size_t K = (A->symmetric || parts < 2) ? 0 : device_plan_chunks(A->dev, (size_t) parts, bounds, 1);
        if (K == 0) {
            device_spmv(A->dev, alpha, x_dev, beta, y_dev, stream);
        } else {
            for (size_t k = 0; k < K; ++k) device_spmv_chunk(A->dev, k, alpha, x_dev, beta, y_dev, stream, 1);
        }
        if (launched) *launched = K ? (int) K : 1;
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_matvec_dist(spx_value_t alpha, const spx_matrix_t *A,
                                const spx_value_t *x_dev, spx_value_t beta,
                                spx_value_t *y_dev, int flags, void *stream)
try {
    if (check_dev(A, x_dev, y_dev) != SPX_SUCCESS) return SPX_FAILURE;
    if (!A->dist) {
        SETERROR_1(SPX_ERR_TUNED_MAT, "matrix has no exchange plan (spx_hip_mat_dist_attach)");
        return SPX_FAILURE;
    }
    try {
        if ((flags & SPX_DIST_OVERLAP) && (flags & SPX_DIST_HALO_X) && !(flags & SPX_DIST_GATHER_Y) &&
            !A->symmetric && A->dist->rounds && A->dist->world > 1) {
            dist_step_overlapped(A->dist, A->dev, alpha, x_dev, beta, y_dev, stream);
            return SPX_SUCCESS;
        }
        device_spmv(A->dev, alpha, x_dev, beta, y_dev, stream);
        dist_complete(A->dist, y_dev, (flags & SPX_DIST_GATHER_Y) != 0, (flags & SPX_DIST_HALO_X) != 0, stream);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    }
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

// ======================================================================================
//  extensions: info / export
// ======================================================================================

int spx_hip_abi_version(void) { return SPX_HIP_ABI_VERSION; }

int spx_hip_mat_host_parts(const spx_matrix_t *A)
try {
    return A && A->dev ? device_host_parts(A->dev) : 0;
} SPX_C_BOUNDARY(return 0;)

int spx_hip_mat_host_order(const spx_matrix_t *A, int32_t *order, int cap)
try {
    return A && A->dev && (order || cap <= 0) ? device_host_order(A->dev, order, cap < 0 ? 0 : cap) : 0;
} SPX_C_BOUNDARY(return 0;)

spx_error_t spx_hip_mat_unit_windows(spx_matrix_t *A, uint32_t budget, uint32_t gap, spx_hip_xw_plan_t *out)
try {
    if (!A || !out) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument"); return SPX_FAILURE; }
    const GpuStream *s = A->host_stream ? A->host_stream.get() : A->index.get();
    if (!s) { SETERROR_1(SPX_ERR_TUNED_MAT, "matrix holds no descriptor stream"); return SPX_FAILURE; }
    std::lock_guard<std::mutex> lk(A->mtx);
    try {
        std::unique_ptr<XwPlan> plan(new XwPlan);
        plan_unit_xwindows(*s, (size_t) A->ncols, A->symmetric ? 0u : budget, gap, *plan, host_threads());
        A->xw_inspect = std::move(plan);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    } catch (const std::exception &e) {
        SETERROR_1(SPX_ERR_MEM_ALLOC, e.what());
        return SPX_FAILURE;
    }
    const XwPlan &p = *A->xw_inspect;
    memset(out, 0, sizeof(*out));
    out->tab = reinterpret_cast<const uint32_t *>(p.tab.data());
    out->xdescs = reinterpret_cast<const uint32_t *>(p.xdescs.data());
    out->passes = p.passes.data();
    out->n_rowblocks = s->rbs.size();
    out->n_descs = p.xdescs.size();
    out->n_passes = p.passes.size();
    out->rowblocks_with_windows = p.n_rb_windows;
    out->rowblocks_with_units = p.n_rb_units;
    out->staged_doubles = p.staged_doubles;
    out->unit_elems = p.unit_elems;
    out->unit_elems_lds = p.unit_elems_lds;
    out->lds_doubles = p.lds_doubles;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

int64_t spx_hip_mat_x_pieces(spx_matrix_t *A, size_t piece, uint64_t *mask, uint32_t *row0, uint32_t *n_rows, size_t cap)
try {
    if (!A || !piece) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument"); return -1; }
    const GpuStream *s = A->host_stream ? A->host_stream.get() : A->index.get();
    if (!s) { SETERROR_1(SPX_ERR_TUNED_MAT, "matrix holds no descriptor stream"); return -1; }
    std::lock_guard<std::mutex> lk(A->mtx);
    std::vector<uint64_t> m;
    stream_rowblock_xpieces(*s, (size_t) A->ncols, piece, m, host_threads());
    for (size_t i = 0; i < m.size() && i < cap; ++i) {
        if (mask) mask[i] = m[i];
        if (row0) row0[i] = s->rbs[i].row0;
        if (n_rows) n_rows[i] = s->rbs[i].n_rows;
    }
    return (int64_t) m.size();
} SPX_C_BOUNDARY(return -1;)

spx_error_t spx_hip_mat_sym_pipeline(spx_matrix_t *A, spx_hip_sx_plan_t *out)
try {
    if (!A || !out) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument"); return SPX_FAILURE; }
    const GpuStream *s = A->host_stream ? A->host_stream.get() : A->index.get();
    if (!s) { SETERROR_1(SPX_ERR_TUNED_MAT, "matrix holds no descriptor stream"); return SPX_FAILURE; }
    std::lock_guard<std::mutex> lk(A->mtx);
    try {
        std::unique_ptr<SxPlan> plan(new SxPlan);
        plan_sym_pipeline(*s, *plan, host_threads());
        A->sx_inspect = std::move(plan);
    } catch (const FatalError &e) {
        SETERROR_1(SPX_ERR_TUNED_MAT, e.what.c_str());
        return SPX_FAILURE;
    } catch (const std::exception &e) {
        SETERROR_1(SPX_ERR_MEM_ALLOC, e.what());
        return SPX_FAILURE;
    }
    const SxPlan &p = *A->sx_inspect;
    memset(out, 0, sizeof(*out));
    out->passes = p.passes.data();
    out->n_sx = p.n_sx.data();
    out->n_rowblocks = s->rbs.size();
    out->n_passes = p.passes.size();
    out->rowblocks_with_sx = p.n_rb_sx;
    out->sym_elems = p.sym_elems;
    out->sx_elems = p.sx_elems;
    out->sym_passes = p.sym_passes;
    out->sx_passes = p.sx_passes;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_mat_info_sized(const spx_matrix_t *A, void *info, size_t size)
try {
    spx_hip_info_t full;
    if (!info || spx_hip_mat_info(A, &full) != SPX_SUCCESS) {
        if (!info) SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument");
        return SPX_FAILURE;
    }
    memcpy(info, &full, std::min(size, sizeof(full)));
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_hip_mat_info(const spx_matrix_t *A, spx_hip_info_t *info)
try {
    if (!A || !info) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument"); return SPX_FAILURE; }
    memset(info, 0, sizeof(*info));
    info->nnz = A->nnz;
    info->nnz_stored = (int64_t) A->nnz_stored;
    info->n_unit_elems = (int64_t) A->n_unit_elems;
    info->n_delta_elems = (int64_t) A->n_delta_elems;
    info->n_units = (int64_t) A->n_units;
    info->n_rowblocks = (int64_t) A->n_rowblocks;
    info->n_shared_rows = (int64_t) A->n_shared;
    info->value_bytes = (int64_t) A->value_bytes;
    info->index_bytes = (int64_t) A->index_bytes;
    info->nr_partitions = (int32_t) A->nr_partitions;
    info->first_partition = (int32_t) A->first_part;
    info->last_partition = (int32_t) A->last_part;
    info->row_lo = A->own_lo;
    info->row_hi = A->own_hi;
    info->symmetric = A->symmetric;
    info->on_device = A->dev ? 1 : 0;
    if (A->dev) {
        DeviceMatrixInfo di;
        device_info(A->dev, di);
        info->device = di.device;
    }
    info->waves = A->dev ? device_get_waves(A->dev) : A->waves;
    info->sym_tiles = A->has_tiles ? ((A->dev ? device_get_sym_atomic(A->dev) : A->sym_atomic) ? 2 : 1) : 0;
    info->wave_tiles = A->dev ? (device_get_wave_tiles(A->dev) ? 1 : 0) : (A->wave_tiles == 1 || A->deterministic ? 1 : 0);
    info->sym_segments = A->has_symsegs ? (A->has_symtiles ? 1 : 2) : 0;
    info->col_slices = A->col_phases > 1 ? (A->col_concurrent ? (int32_t) A->col_phases : -(int32_t) A->col_phases) : 1;
    info->quad = 0;                  // (reserved: the four-pass kernel variant of round 3 is gone)
    if (A->dev && device_has_xw(A->dev)) {
        uint64_t el = 0, ue = 0, st = 0;
        uint32_t lds = 0;
        device_xw_info(A->dev, el, ue, st, lds);
        info->unit_windows = device_get_xw(A->dev) ? 1 : 0;
        info->unit_window_lds = (int32_t) lds;
        info->unit_window_elems = (int64_t) el;
        info->unit_window_staged = (int64_t) st;
    }
    if (A->dev && device_has_sx(A->dev)) {
        uint64_t esx = 0, esym = 0;
        size_t nrb = 0;
        device_sx_info(A->dev, esx, esym, nrb);
        info->sym_pipeline = device_get_sx(A->dev) ? 1 : 0;
        info->sym_pipeline_elems = (int64_t) esx;
    }
    info->tune_seconds = A->tune_seconds;
    info->emit_seconds = A->emit_seconds;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

static bool owned_part(const spx_matrix_t *A, int part)
{
    return part >= (int) A->first_part && part < (int) A->last_part &&
           (size_t)(part - (int) A->first_part) < A->parts.size();
}

spx_error_t spx_hip_mat_export_csx(const spx_matrix_t *A_, int part, spx_csx_export_t *out)
try {
    spx_matrix_t *A = const_cast<spx_matrix_t *>(A_);
    if (!A || !out) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid argument"); return SPX_FAILURE; }
    if (!owned_part(A, part)) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "partition not held by this process "
                   "(or dropped: spx.rt.keep_encoded=false)");
        return SPX_FAILURE;
    }
    std::lock_guard<std::mutex> lk(A->mtx);
    size_t li = (size_t) part - A->first_part;
    if (A->exported.size() < A->parts.size()) {
        A->exported.resize(A->parts.size());
        A->exported_rows_info.resize(A->parts.size());
    }
    if (!A->exported[li]) {
        std::unique_ptr<CsxStream> s(new CsxStream);
        emit_csx(A->parts[li], A->full_colind, A->symmetric != 0, *s);
        if (A->symmetric) {
            s->dvalues = A->diag[li];
            s->dvalues.resize((size_t) s->nrows, 0.0);
        }
        std::vector<spx_index_t> &ri = A->exported_rows_info[li];
        ri.resize(s->rows_info.size() * 3);
        for (size_t i = 0; i < s->rows_info.size(); ++i) {
            ri[3 * i] = s->rows_info[i].rowptr;
            ri[3 * i + 1] = s->rows_info[i].valptr;
            ri[3 * i + 2] = s->rows_info[i].span;
        }
        A->exported[li] = std::move(s);
    }
    const CsxStream &s = *A->exported[li];
    memset(out, 0, sizeof(*out));
    out->values = s.values.data();
    out->ctl = s.ctl.data();
    out->ctl_size = (int64_t) s.ctl.size();
    out->nnz = s.nnz;
    out->ncols = s.ncols;
    out->nrows = s.nrows;
    out->row_start = s.row_start;
    out->row_jumps = s.row_jumps;
    out->full_colind = s.full_colind;
    for (int i = 0; i < 64; ++i) out->id_map[i] = s.id_map[i];
    out->rows_info = A->exported_rows_info[li].data();
    out->dvalues = A->symmetric ? s.dvalues.data() : NULL;
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

int64_t spx_hip_mat_export_units(const spx_matrix_t *A, int part, spx_unit_record_t *recs,
                                 int64_t cap)
try {
    if (!A || !owned_part(A, part)) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "partition not held by this process");
        return -1;
    }
    const Partition &p = A->parts[(size_t) part - A->first_part];
    for (size_t i = 0; i < p.elems_size && (int64_t) i < cap && recs; ++i) {
        const Elem &e = p.elems[i];
        recs[i].type = e.type;
        recs[i].delta = (int32_t) e.delta;
        recs[i].size = e.size;
        recs[i].row = e.row;
        recs[i].col = e.col;
    }
    return (int64_t) p.elems_size;
} SPX_C_BOUNDARY(return 0;)

const char *spx_hip_mat_tune_log(const spx_matrix_t *A)
try {
    return A ? A->log.c_str() : "";
} SPX_C_BOUNDARY(return nullptr;)

// ======================================================================================
//  vectors (host side; reference src/internals/Vector.cpp)
// ======================================================================================

static spx_vector_t *vec_alloc(size_t size)
{
    spx_vector_t *v = (spx_vector_t *) malloc(sizeof(spx_vector_t));
    if (!v) { log_msg(LOG_ERR, "malloc failed\n"); exit(1); }
    // page-locked where a HIP device is present: spx_matvec_* then copy the
    // vector to and from HBM without a staging copy
    v->alloc_type = ALLOC_PINNED;
    v->elements = (spx_value_t *) device_host_alloc((size ? size : 1) * sizeof(spx_value_t));
    if (v->elements) {
        memset(v->elements, 0, (size ? size : 1) * sizeof(spx_value_t));
    } else {
        v->elements = (spx_value_t *) calloc(size ? size : 1, sizeof(spx_value_t));
        v->alloc_type = ALLOC_STD;
    }
    if (!v->elements) { log_msg(LOG_ERR, "malloc failed\n"); exit(1); }
    v->size = size;
    v->vec_mode = VEC_MODE_INVALID;
    vec_track(v);
    return v;
}

spx_vector_t *spx_vec_create(size_t size, const spx_partition_t *p)
try {
    if (p == SPX_INVALID_PART) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid partition handle");
        return SPX_INVALID_VEC;
    }
    return vec_alloc(size);
} SPX_C_BOUNDARY(return SPX_INVALID_VEC;)

spx_vector_t *spx_vec_create_from_buff(spx_value_t *buff, spx_value_t **tuned, size_t size,
                                       const spx_partition_t *p, spx_vecmode_t mode)
try {
    if (!buff) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid buffer"); return SPX_INVALID_VEC; }
    if (!check_vecmode(mode)) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector mode");
        return SPX_INVALID_VEC;
    }
    if (p == SPX_INVALID_PART && mode == SPX_VEC_TUNE) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid partition handle");
        return SPX_INVALID_VEC;
    }
    // both modes alias the user buffer, as on the reference's non-NUMA build (src/api/matvec.c:809-815): the
    // client writes x through its own pointer, so such a vector is NEVER tracked for residency (no vec_track)
    spx_vector_t *v = (spx_vector_t *) malloc(sizeof(spx_vector_t));
    if (!v) { log_msg(LOG_ERR, "malloc failed\n"); exit(1); }
    v->elements = buff;
    v->size = size;
    v->alloc_type = ALLOC_OTHER;
    v->vec_mode = (int) mode;
    if (tuned) *tuned = buff;
    return v;
} SPX_C_BOUNDARY(return SPX_INVALID_VEC;)

void spx_vec_init_rand_range(spx_vector_t *v, spx_value_t max, spx_value_t min)
try {
    vec_touch(v);
    for (size_t i = 0; i < v->size; i++) {
        spx_value_t val = ((spx_value_t)(rand() + i) / ((spx_value_t) RAND_MAX + 1));
        v->elements[i] = min + val * (max - min);
    }
} SPX_C_BOUNDARY(return;)

void spx_hip_vec_touch(const spx_vector_t *v)
try {
    if (v) vec_touch(v);
} SPX_C_BOUNDARY(return;)

spx_vector_t *spx_vec_create_random(size_t size, const spx_partition_t *p)
try {
    if (p == SPX_INVALID_PART) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid partition handle");
        return SPX_INVALID_VEC;
    }
    spx_vector_t *v = vec_alloc(size);
    // values in (-0.1, 0.1], same argument order as Vector.cpp:161-167
    spx_vec_init_rand_range(v, (spx_value_t) -0.1, (spx_value_t) 0.1);
    return v;
} SPX_C_BOUNDARY(return SPX_INVALID_VEC;)

void spx_vec_init(spx_vector_t *v, spx_value_t val)
try {
    vec_touch(v);
    for (size_t i = 0; i < v->size; i++) v->elements[i] = val;
} SPX_C_BOUNDARY(return;)

void spx_vec_init_part(spx_vector_t *v, spx_value_t val, spx_index_t start, spx_index_t end)
try {
    vec_touch(v);
    for (spx_index_t i = start; i < end; i++) v->elements[i] = val;
} SPX_C_BOUNDARY(return;)

spx_error_t spx_vec_set_entry(spx_vector_t *v, spx_index_t idx, spx_value_t val, ...)
try {
    va_list ap;
    va_start(ap, val);
    spx_option_t indexing = va_arg(ap, spx_option_t);
    va_end(ap);
    bool one_based = (indexing == SPX_INDEX_ONE_BASED);
    long pos = (long) idx - (one_based ? 1 : 0);
    if (!v || pos < 0 || (size_t) pos >= v->size) {
        SETERROR_0(SPX_OUT_OF_BOUNDS);
        SETWARNING(SPX_WARN_ENTRY_NOT_SET);
        return SPX_FAILURE;
    }
    v->elements[pos] = val;
    vec_touch(v);
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

void spx_vec_scale(spx_vector_t *v1, spx_vector_t *v2, spx_value_t num)
try {
    vec_touch(v2);
    for (size_t i = 0; i < v1->size; i++) v2->elements[i] = num * v1->elements[i];
} SPX_C_BOUNDARY(return;)

void spx_vec_scale_add(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3, spx_value_t num)
try {
    vec_touch(v3);
    for (size_t i = 0; i < v1->size; i++)
        v3->elements[i] = v1->elements[i] + num * v2->elements[i];
} SPX_C_BOUNDARY(return;)

void spx_vec_scale_add_part(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3,
                            spx_value_t num, spx_index_t start, spx_index_t end)
try {
    vec_touch(v3);
    for (spx_index_t i = start; i < end; i++)
        v3->elements[i] = v1->elements[i] + num * v2->elements[i];
} SPX_C_BOUNDARY(return;)

void spx_vec_add(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3)
try {
    vec_touch(v3);
    if (v1->size != v2->size || v1->size != v3->size) {
        fprintf(stderr, "v1->size=%lu v2->size=%lu v3->size=%lu differ\n",
                (unsigned long) v1->size, (unsigned long) v2->size, (unsigned long) v3->size);
        exit(1);
    }
    for (size_t i = 0; i < v1->size; i++) v3->elements[i] = v1->elements[i] + v2->elements[i];
} SPX_C_BOUNDARY(return;)

void spx_vec_add_part(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3,
                      spx_index_t start, spx_index_t end)
try {
    vec_touch(v3);
    for (spx_index_t i = start; i < end; i++)
        v3->elements[i] = v1->elements[i] + v2->elements[i];
} SPX_C_BOUNDARY(return;)

void spx_vec_sub(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3)
try {
    vec_touch(v3);
    for (size_t i = 0; i < v1->size; i++) v3->elements[i] = v1->elements[i] - v2->elements[i];
} SPX_C_BOUNDARY(return;)

void spx_vec_sub_part(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3,
                      spx_index_t start, spx_index_t end)
try {
    vec_touch(v3);
    for (spx_index_t i = start; i < end; i++)
        v3->elements[i] = v1->elements[i] - v2->elements[i];
} SPX_C_BOUNDARY(return;)

spx_value_t spx_vec_mul(const spx_vector_t *v1, const spx_vector_t *v2)
try {
    spx_value_t ret = 0;
    for (size_t i = 0; i < v1->size; i++) ret += v1->elements[i] * v2->elements[i];
    return ret;
} SPX_C_BOUNDARY(return 0;)

spx_value_t spx_vec_mul_part(const spx_vector_t *v1, const spx_vector_t *v2,
                             spx_index_t start, spx_index_t end)
try {
    spx_value_t ret = 0;
    for (spx_index_t i = start; i < end; i++) ret += v1->elements[i] * v2->elements[i];
    return ret;
} SPX_C_BOUNDARY(return 0;)

spx_error_t spx_vec_reorder(spx_vector_t *v, spx_perm_t *p)
try {
    if (p == SPX_INVALID_PERM) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid permutation");
        return SPX_FAILURE;
    }
    std::vector<spx_value_t> tmp(v->size);
    for (size_t i = 0; i < v->size; i++) tmp[(size_t) p[i]] = v->elements[i];
    memcpy(v->elements, tmp.data(), v->size * sizeof(spx_value_t));
    vec_touch(v);
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

spx_error_t spx_vec_inv_reorder(spx_vector_t *v, spx_perm_t *p)
try {
    if (p == SPX_INVALID_PERM) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "invalid permutation");
        return SPX_FAILURE;
    }
    std::vector<spx_value_t> tmp(v->size);
    for (size_t i = 0; i < v->size; i++) tmp[i] = v->elements[(size_t) p[i]];
    memcpy(v->elements, tmp.data(), v->size * sizeof(spx_value_t));
    vec_touch(v);
    return SPX_SUCCESS;
} SPX_C_BOUNDARY(return SPX_FAILURE;)

void spx_vec_copy(const spx_vector_t *v1, spx_vector_t *v2)
try {
    vec_touch(v2);
    memcpy(v2->elements, v1->elements, v1->size * sizeof(spx_value_t));
} SPX_C_BOUNDARY(return;)

int spx_vec_compare(const spx_vector_t *v1, const spx_vector_t *v2)
try {
    // relative tolerance 1e-6 per element (Vector.cpp:51-57, :396-413)
    if (v1->size != v2->size) {
        fprintf(stderr, "v1->size=%lu v2->size=%lu differ\n", (unsigned long) v1->size,
                (unsigned long) v2->size);
        return -2;
    }
    for (size_t i = 0; i < v1->size; i++) {
        double a = v1->elements[i], b = v2->elements[i];
        if (fabs((a - b) / a) > 1.e-6) {
            fprintf(stderr, "element %ld differs: %10.20f != %10.20f\n", (long) i, a, b);
            return -1;
        }
    }
    return 0;
} SPX_C_BOUNDARY(return 0;)

void spx_vec_print(const spx_vector_t *v)
try {
    printf("[ ");
    for (size_t i = 0; i < v->size; i++) printf("%g ", v->elements[i]);
    printf("]\n");
} SPX_C_BOUNDARY(return;)

int spx_hip_vec_page_locked(const spx_vector_t *v)
try {
    if (!v) return 0;
    if (v->alloc_type == ALLOC_PINNED) return 1;
    std::lock_guard<std::mutex> lk(g_vec_mtx);
    auto it = g_vec_reg.find(v);
    return it != g_vec_reg.end() && it->second.locked && it->second.ptr == (void *) v->elements ? (it->second.ours ? 2 : 3) : 0;
} SPX_C_BOUNDARY(return 0;)

void spx_vec_destroy(spx_vector_t *v)
try {
    if (!v) return;
    vec_forget(v);
    vec_release_lock(v);
    if (v->alloc_type == ALLOC_STD) free(v->elements);
    else if (v->alloc_type == ALLOC_PINNED) device_host_free(v->elements);
    free(v);
} SPX_C_BOUNDARY(return;)

}  // extern "C"
