// dist.hpp -- row-partitioned matrices, one process per GPU: the exchange plan
// that completes y and its device side (see include/sparsex_hip.h, "one process
// per GPU").  Counterpart of the reference's conflict map and its reduction
// (include/sparsex/internals/CsxBuild.hpp:400-581, Map.hpp:23-27,
// src/internals/Vector.cpp:291-299, src/internals/CsxSpmv.cpp:37-50).
#pragma once

#include <sparsex_hip.h>

#include "common.hpp"
#include "device.hpp"

#include <vector>

namespace spx {

struct DistDevice;     // device arrays of a plan (dist_kernels.hip)

constexpr size_t DIST_MAX_CHUNKS = 64;     // parts of the own product in the overlapped step (spx.rt.dist_chunks is clamped to it)

struct DistPlan {
    spx_hip_transport_t transport;
    int rank = 0, world = 1;
    std::vector<idx_t> row_lo, row_hi;            // every process' rows
    std::vector<idx_t> send_rows;                 // this process' conflict rows (ascending)
    std::vector<size_t> send_off, send_cnt;       // per owner
    std::vector<size_t> recv_off, recv_cnt;       // per sender
    size_t n_recv = 0;
    std::vector<idx_t> fix_rows;                  // own rows that receive something
    std::vector<uint32_t> fix_ptr, fix_pos;       // per such row: positions in the receive buffer
    bool any_exchange = false;                    // some process sends something
    std::vector<size_t> gat_send_off, gat_send_cnt, gat_recv_off, gat_recv_cnt;   // slices of y, in place
    // the halo of x (SPX_DIST_HALO_X): instead of handing whole slices of y round, every process
    // receives exactly the entries its own rows read as x
    std::vector<idx_t> halo_cols;                 // entries this process needs (ascending: grouped by owner)
    std::vector<size_t> halo_recv_off, halo_recv_cnt;   // per owner
    std::vector<idx_t> halo_send_rows;            // own entries the others need, grouped by the process that asked
    std::vector<size_t> halo_send_off, halo_send_cnt;   // per such process
    // overlap (SPX_DIST_OVERLAP): the own product runs in `my_chunks` launches over consecutive parts of
    // the rows; the halo entries of part k travel (round k, on a second stream) while part k + 1 is
    // computed.  Every process takes part in `rounds` = the largest part count of any process.
    size_t my_chunks = 0, rounds = 0;
    std::vector<size_t> chunk_rows;                               // my_chunks + 1 row bounds
    std::vector<std::vector<size_t>> rd_send_off, rd_send_cnt;    // [round][peer] inside the halo send buffer
    std::vector<std::vector<size_t>> rd_recv_off, rd_recv_cnt;    // [round][peer] inside the halo receive buffer
    std::vector<uint32_t> rd_pack_pos, rd_scat_pos;               // positions to pack / scatter, round by round
    std::vector<size_t> rd_pack_ptr, rd_scat_ptr;                 // rounds + 1 offsets into them
    DistDevice *dev = nullptr;
};

// Collective.  `conflict_rows`: rows in front of own_lo this process adds to.
// Throws FatalError.
// `halo_cols`: entries of x outside the own rows that this process' products read.
DistPlan *dist_build_plan(const spx_hip_transport_t &t, idx_t own_lo, idx_t own_hi, idx_t nrows,
                          const std::vector<idx_t> &conflict_rows, const std::vector<idx_t> &halo_cols,
                          bool on_device);
void dist_free_plan(DistPlan *p);

// Collective, after dist_build_plan: agree on the rounds of the overlapped step.  `chunk_rows`: the
// row bounds of this process' parts (empty: its stream cannot be cut -- it sends everything in round 0).
void dist_plan_overlap(DistPlan *p, const std::vector<size_t> &chunk_rows);

// the overlapped step of the general path: part k of the product on `stream`, round k of the halo
// exchange behind it on the plan's second stream; returns with `stream` waiting for the last round
void dist_step_overlapped(DistPlan *p, DeviceMatrix *m, double alpha, const double *d_x, double beta,
                          double *d_y, void *stream);

// after the local SpMV on `stream`: hand the sums for other processes' rows to
// their owners and add what arrives; then (gather) pass the finished slices round
// ... or (halo) send every process the entries of the own slice that its rows read as x
void dist_complete(DistPlan *p, double *d_y, bool gather, bool halo, void *stream);

// device side (dist_kernels.hip)
DistDevice *dist_device_create(const DistPlan &p);
void dist_device_free(DistDevice *d);
const double *dist_device_pack(DistDevice *d, const double *d_y, void *stream);
double *dist_device_recvbuf(DistDevice *d);
void dist_device_unpack(DistDevice *d, double *d_y, void *stream);
const double *dist_device_halo_pack(DistDevice *d, const double *d_y, void *stream);
double *dist_device_halo_recvbuf(DistDevice *d);
void dist_device_halo_scatter(DistDevice *d, double *d_y, void *stream);
// overlapped step: round lists on the device, the second stream and its events
void dist_device_set_rounds(DistDevice *d, const DistPlan &p);
void *dist_device_comm_stream(DistDevice *d);
void dist_device_part_done(DistDevice *d, size_t part, void *main_stream);     // records the part's event on the launch stream
void dist_device_round_begin(DistDevice *d, size_t round);                      // the second stream waits for that part
void dist_device_round_pack(DistDevice *d, size_t round, const double *d_y);
void dist_device_round_scatter(DistDevice *d, size_t round, double *d_y);
void dist_device_rounds_end(DistDevice *d, void *main_stream);                    // main stream waits for the last round
double *dist_device_halo_sendbuf(DistDevice *d);

}  // namespace spx
