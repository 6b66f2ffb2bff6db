// common.hpp -- shared host-side vocabulary of the CSX preprocessor.
//
// Encoding types follow the reference's Encoding::Type numbering
// (include/sparsex/internals/Encodings.hpp:38-66) because the numeric value
// is part of the on-"wire" pattern id (type*10000 + delta,
// include/sparsex/internals/CsxUtil.hpp:58-74).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include <utility>

namespace spx {

typedef int idx_t;       // spx_index_t
typedef double val_t;    // spx_value_t

enum EncType : int {
    ENC_NONE = 0,
    ENC_H = 1,
    ENC_V = 2,
    ENC_D = 3,
    ENC_AD = 4,
    ENC_BR1 = 5,   // .. ENC_BR8 = 12
    ENC_BR8 = 12,
    ENC_BC1 = 13,  // .. ENC_BC8 = 20
    ENC_BC8 = 20,
    ENC_MAX = 21,  // __EndOfTypes__
    // groups (only valid in option strings)
    ENC_GROUP_BR = 22,
    ENC_GROUP_BC = 23,
    ENC_GROUP_ALL = 24
};

inline bool enc_is_block_row(int t) { return t >= ENC_BR1 && t <= ENC_BR8; }
inline bool enc_is_block_col(int t) { return t >= ENC_BC1 && t <= ENC_BC8; }
inline bool enc_is_block(int t) { return enc_is_block_row(t) || enc_is_block_col(t); }
// fixed ("aligned") dimension of a block type, 0 for linear types
inline int enc_block_align(int t)
{
    if (enc_is_block_row(t)) return t - ENC_BR1 + 1;
    if (enc_is_block_col(t)) return t - ENC_BC1 + 1;
    return 0;
}

const char *enc_short_name(int t);
const char *enc_full_name(int t);
// returns -1 when the short name is unknown
int enc_from_short_name(const std::string &s);
// expands a type or group into concrete types
void enc_expand(int t, std::vector<int> &out);

const unsigned long PATTERN_ID_OFFSET = 10000;  // CsxUtil.cpp:27
const int CTL_PATTERNS_MAX = 63;                // CtlUtil.hpp:59
const int CTL_SIZE_MAX = 255;                   // CtlUtil.hpp:62

// logging --------------------------------------------------------------
enum LogLevel { LOG_NONE = 0, LOG_ERR = 1, LOG_WARN = 2, LOG_INFO = 3,
                LOG_VERB = 4, LOG_DBG = 5 };
void log_set_level(int level);
void log_set_file(const char *path);   // NULL -> stderr
void log_msg(int level, const char *fmt, ...);

// thrown for malformed input; the C API turns it into exit(1) where the
// reference exits (Mmf.hpp:259-263, SparseInternal.hpp:147-151)
struct FatalError {
    std::string what;
    explicit FatalError(const std::string &w) : what(w) {}
};

}  // namespace spx
