/* sparsex/config.h -- build-time type configuration (reference: config.h.in). */
#ifndef SPARSEX_CONFIG_H
#define SPARSEX_CONFIG_H

#define SPX_INDEX_TYPE  int
#define SPX_VALUE_TYPE  double

/* This build drives an AMD MI355X (gfx950) through HIP. */
#define SPX_BACKEND_HIP_GFX950 1

#endif /* SPARSEX_CONFIG_H */
