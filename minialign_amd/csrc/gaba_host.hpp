/* gaba_host.hpp -- host-side internals shared by gaba_batch.hip and mm_host.hip */
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "gaba_device.hpp"
#include "../../include/gaba.h"

#define HIP_OK(_e, _ret) do { hipError_t _r = (_e); if(_r != hipSuccess) { \
	fprintf(stderr, "[minialign_amd] HIP error %s at %s:%d\n", hipGetErrorString(_r), __FILE__, __LINE__); return _ret; } } while(0)

struct gaba_arena_s { uint32_t *pk, *nm; uint64_t n; const uint8_t *host; };     /* host: the array it was uploaded from (section lookup of the per-call API) */

/* drop an arena from the host-range registry of the per-call API (its host array is about to go away); not part of the C ABI */
extern "C" void gaba_arena_unregister(gaba_arena_t *ar);

struct gaba_context_s {
	gaba::Consts hc;
	gaba::Consts *dc;
	uint8_t *droots;              /* 3 x (Blk + Tail) */
	uint8_t *slabs; uint64_t slab_bytes; uint32_t n_waves;
	uint32_t *counter; uint64_t *dstats;
	hipStream_t stream;
	hipEvent_t ev0, ev1;
	gaba_batch_stats_t last;
};

