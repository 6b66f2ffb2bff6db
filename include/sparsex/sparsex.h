/* sparsex/sparsex.h -- umbrella header (reference: include/sparsex/sparsex.h). */
#ifndef SPARSEX_SPARSEX_H
#define SPARSEX_SPARSEX_H

#include <sparsex/common.h>
#include <sparsex/error.h>
#include <sparsex/matvec.h>
#include <sparsex/timing.h>
#include <sparsex/types.h>

#endif /* SPARSEX_SPARSEX_H */
