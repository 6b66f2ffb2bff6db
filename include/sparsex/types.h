/*
 * sparsex/types.h -- index and value types of the MI355X-native CSX SpMV library.
 *
 * Mirrors the type choices of the reference (include/sparsex/types.h:25-35,
 * configure.ac:87-111): 32-bit signed indices, IEEE double values.
 */
#ifndef SPARSEX_TYPES_H
#define SPARSEX_TYPES_H

#include <sparsex/config.h>

typedef SPX_INDEX_TYPE spx_index_t;
typedef SPX_VALUE_TYPE spx_value_t;

#endif /* SPARSEX_TYPES_H */
