// handles.hpp -- what the opaque handles of <sparsex/sparsex.h> point to
// (reference: the structs at the top of src/api/matvec.c:30-62).
#pragma once

#include <sparsex/sparsex.h>
#include <sparsex_hip.h>

#include "csx_emit.hpp"
#include "device.hpp"
#include "gpu_emit.hpp"
#include "input.hpp"
#include "xwindows.hpp"
#include "sxplan.hpp"

#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <thread>

namespace spx { struct DistPlan; }

using namespace spx;

struct input {
    spx_index_t nrows, ncols, nnz;
    char type;                         // 'C' (CSR) or 'M' (MMF)
    MatrixInput *mat;
};

struct partition {
    size_t nr_partitions;
    size_t *parts;
    int *nodes;
    int *affinity;
    spx_index_t *row_start;
    spx_index_t *row_end;
};

struct matrix {
    spx_index_t nrows = 0, ncols = 0, nnz = 0;
    int symmetric = 0;
    spx_perm_t *permutation = SPX_INVALID_PERM;
    // tuned representation
    size_t nr_partitions = 0;          // P, over all processes
    size_t first_part = 0, last_part = 0;   // owned partitions [first, last)
    std::vector<PartBounds> bounds;    // all P partitions
    std::vector<Partition> parts;      // encoded, horizontal order (owned ones)
    // Large intermediates of the tune -- the encoded partitions with spx.rt.keep_encoded=false, the symmetric
    // path's expanded ranges -- are handed back behind the caller's back (the kernel clears what it takes
    // back: 1.6 s for the 25 GB of the contract matrix on sixteen threads, and nothing waits for it); the
    // threads are joined when the matrix goes
    std::vector<std::thread> release_threads;
    // (a thread that cannot be started does the work here and now: the closure must run, it owns the memory)
    template <class F> void release_later(F &&f)
    {
        std::function<void()> work(std::forward<F>(f));
        try {
            release_threads.emplace_back([work] { work(); });
        } catch (const std::exception &) {
            work();
        }
    }
    // waits for what is still being handed back: before anything is timed (the threads keep every host CPU
    // and the kernel's mmap lock busy), before a new release starts, and when the matrix goes
    void release_wait()
    {
        for (std::thread &t : release_threads)
            if (t.joinable()) t.join();
        release_threads.clear();
    }
    ~matrix() { release_wait(); }
    std::vector<std::vector<val_t>> diag;   // symmetric: per owned partition
    std::vector<std::unique_ptr<CsxStream>> exported;
    std::vector<std::vector<spx_index_t>> exported_rows_info;
    bool full_colind = false;
    DeviceMatrix *dev = nullptr;
    std::unique_ptr<GpuStream> host_stream;   // kept for host-only matrices (save/restore)
    std::unique_ptr<GpuStream> index;         // matrix in HBM: host copy of the stream's index arrays (no values),
                                              // walked by get/set entry
    idx_t own_lo = 0, own_hi = 0;
    // symmetric matrix of which this process holds a slice: the rows in front of
    // its own that it adds to (the reference's conflict map, CsxBuild.hpp:400-451)
    std::vector<idx_t> conflict_rows;
    // columns of x outside the own rows that this process' stream reads (stream_read_cols): what it
    // needs of the other processes' slices of a vector -- the halo of SPX_DIST_HALO_X
    std::vector<idx_t> halo_cols;
    idx_t first_block_row = 0;                // first row a row-block of this process covers (<= own_lo)
    spx::DistPlan *dist = nullptr;            // set by spx_hip_mat_dist_attach
    GpuEmitParams emit_params;
    bool auto_rb = false;
    double rb_scale = 1.0;      // chosen by the launch autotuner (multiplies the automatic row-block size)
    int waves = 4;              // wavefronts per workgroup of the SpMV kernel
    size_t col_phases = 1;      // general path: column slices the stream is emitted in (spx.gpu.col_phases)
    bool col_concurrent = false;   // ... all of them in one launch, a group of XCDs each (SPX_RB_ACCUM)
    bool host_only = false;
    bool has_tiles = false;     // the stream holds SPX_PASS_SYMTILE passes
    bool sym_atomic = false;    // their transposed sums go straight into y (global atomics)
    bool deterministic = false; // spx.gpu.deterministic
    bool has_symtiles = false;  // the stream holds dense 8x8 tiles (SPX_PASS_SYMTILE)
    bool has_symsegs = false;   // the stream holds read-once row segments (SPX_PASS_SYMSEG): atomic hand-over only
    int spill_mode = -1;        // spx.gpu.sym_spill as asked for: 0 lists, 1 atomic, -1 auto
    int wave_tiles = -1;        // per-wavefront y tiles: 1 / 0, -1 = measured at tune time (spx.gpu.wave_tiles)
    int unit_windows = -1;      // spx.gpu.unit_windows: 1 / 0, -1 = measured at tune time
    bool xw_on = false;         // ... the product runs with the unit windows of x in LDS (csx_spmv_xw_kernel)
    bool rb_joined = false;     // general path: row-blocks joined side by side (the launch tuner's choice for streams far beyond the Infinity Cache)
    int sym_pipeline = -1;      // spx.gpu.sym_pipeline: 1 / 0, -1 = measured at tune time
    bool sx_on = false;         // ... the read-once passes run pipelined (csx_spmv_sx_kernel)
    uint32_t xw_budget = 3072, xw_gap = 16;   // spx.gpu.unit_window_doubles, spx.gpu.unit_window_gap
    std::unique_ptr<spx::XwPlan> xw_inspect;    // what spx_hip_mat_unit_windows handed out last
    std::unique_ptr<spx::SxPlan> sx_inspect;    // ... and spx_hip_mat_sym_pipeline
    int device_ordinal = -1;
    std::vector<std::vector<idx_t>> spans;    // per partition and row: reach of its units
    std::vector<idx_t> max_span;              // per partition
    // accounting
    size_t nnz_stored = 0, n_unit_elems = 0, n_delta_elems = 0, n_units = 0;
    size_t value_bytes = 0, index_bytes = 0, n_rowblocks = 0, n_shared = 0;
    double tune_seconds = 0.0, emit_seconds = 0.0;
    std::string log;
    std::mutex mtx;
};

