// csx_emit.hpp -- emitter of the reference-format CSX byte stream.
//
// Produces, for one encoded partition, exactly the arrays the reference's
// CsxManager::MakeCsx / MakeCsxSym produce (include/sparsex/internals/
// CsxManager.hpp:238-706, src/internals/CtlBuilder.cpp:32-93): the `ctl`
// byte stream, the packed `values`, `rows_info[]` and the -1 terminated
// `id_map[]`.  The GPU does not read this stream; it exists so that a tuned
// matrix can be handed to the reference's own SpMV code (the parity oracle
// does that) and for the future save/restore path.
#pragma once

#include "partition.hpp"

#include <cstdint>
#include <vector>

namespace spx {

struct RowInfo { idx_t rowptr, valptr, span; };   // Csx.hpp:29-35

struct CsxStream {
    std::vector<val_t> values;
    std::vector<uint8_t> ctl;
    idx_t nnz = 0, ncols = 0, nrows = 0;
    idx_t row_start = 0;
    bool row_jumps = false;
    bool full_colind = false;
    long id_map[CTL_PATTERNS_MAX + 1];   // slot -> pattern id, -1 terminated
    std::vector<RowInfo> rows_info;
    std::vector<val_t> dvalues;          // symmetric only
};

// `p` must be in horizontal order.  `sym_split` > 0 requests the symmetric
// row walk, which closes the pending delta unit at column `sym_split`
// (= row_start, CsxManager.hpp:559) so that no unit straddles it.
void emit_csx(const Partition &p, bool full_colind, bool symmetric,
              CsxStream &out);

// pattern id of a unit: type*10000 + delta (blocks: the free dimension)
unsigned long unit_pattern_id(const Elem &e);

}  // namespace spx
