// lrb_device.h -- shared between the HIP translation units of liblrb_hip.so: the context,
// the error macros and the grow-on-demand workspace.
#ifndef LRB_DEVICE_H
#define LRB_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lrb_hip.h"
#include "lrb_internal.h"

struct lrb_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int n_cu;
    uint16_t *d_lut[6]; // canonical LUT per k (3..5), device copy
    uint32_t dim[6];
    // workspace slots (grown on demand): 0..7 host-pointer paths, 8..11 window lists (lists, level-1 scratch,
    // unit counts, bounds), 12..15 concatenated batches / HDBSCAN, 16 the sweep's packed map, 17 K4's per-chunk histograms
#define LRB_WS_SLOTS 20
    void *ws[LRB_WS_SLOTS];
    uint64_t ws_bytes[LRB_WS_SLOTS];
    // bumped whenever one of the workspace slots slice lists may live in (8, 11..15) is handed out: lists made in the
    // workspace (lrb_packed_lists_create, in_workspace) are valid while the number they saw still stands
    uint64_t lists_epoch;
    // memory of slice lists of their own (lrb_packed_lists_create, in_workspace = 0) given back by lrb_winlists_free and
    // kept for the next lists instead of going through hipFree / hipMalloc again -- for long-lived hosts that keep
    // lists across the collective call after call (lrb_ctx_list_pool sets the ceiling; 0 = nothing is retained)
#define LRB_POOL_SLOTS 128
    void *pool_ptr[LRB_POOL_SLOTS];
    uint64_t pool_size[LRB_POOL_SLOTS];
    uint64_t pool_cap, pool_held;
    // the slice lists the LAST group of lrb_packed_k15_tally_half_many left in the workspace (round 6): the coverage
    // stage sweeps that group as it stands instead of partitioning its windows again (lrb_packed_cov_hist_many finds it
    // by the batches it was made from; valid while lists_epoch stands and none of those batches has been freed)
    struct lrb_winlists *res_lists;
    const struct lrb_packed **res_packs;
    uint64_t res_count;
    // page-locked staging for the small tables of batches that go to the device ahead of a many-batch launch
    // (lrb_stage_upload: stream-ordered, no synchronisation; the event says when the staging may be written again)
    // partitions of window lists repeated because the count and the part kernel disagreed (lrb_k15_lists_part_dev)
    uint64_t wl_retries;
    void *h_stage;
    uint64_t h_stage_bytes;
    hipEvent_t stage_ev;
    bool stage_ev_live;
};

// (lrb_kernels.hip) `bytes` of host data into workspace slot `slot` behind the work already on the context's stream
int lrb_stage_upload(lrb_ctx *c, int slot, const void *src, uint64_t bytes, void **d_ptr);

// (lrb_kernels.hip) forget the lists kept in the workspace
void lrb_resident_lists_drop(lrb_ctx *c);
void lrb_resident_lists_forget_batch(lrb_ctx *c, const struct lrb_packed *p);

#define HIP_TRY(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            lrb_set_error("%s failed: %s", #call, hipGetErrorString(e_));          \
            return e_ == hipErrorOutOfMemory ? LRB_ERR_NOMEM : LRB_ERR_HIP;        \
        }                                                                          \
    } while (0)

#define ARG_TRY(cond)                                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            lrb_set_error("invalid argument: %s%s", #cond, "");                    \
            return LRB_ERR_ARG;                                                    \
        }                                                                          \
    } while (0)

// Every workgroup barrier of this library: wait for the wave's own outstanding LDS operations, THEN the barrier.
// hipcc leaves LDS out of the fence of __syncthreads() on gfx950 (the waves of a workgroup share one CU, whose LDS executes
// DS instructions in issue order, so a ds_add issued before s_barrier is "ordered" before a ds_read issued after it) -- and
// with several processes time-sliced on one GPU that order did not hold: the count kernel of the window lists read, about once
// in 10^10 tallies, counters that a wave's last ds_add had not reached yet (a count one short, the next unit's one long, a
// window lost two kernels later; found by the eight-rank rehearsal of round 6, profiles/r06_k2_stress.txt: 32 of 1,920
// partitions repeated by the count / part check before this wait, 0 of 1,920 after).  The wait costs nothing where the
// compiler had one anyway or nothing is in flight; it is a correctness requirement wherever LDS atomics or stores of one
// wave are read by another behind the barrier -- i.e. everywhere.
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void lrb_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): vmcnt and expcnt untouched
    __syncthreads();
}
#endif

// hipFuncSetAttribute is per DEVICE: a launch site remembers which devices it has raised its kernel's limit on
// (the C ABI allows contexts on several GPUs in one process).  A race between two threads sets the attribute twice.
struct lrb_per_device_once {
    bool done[64] = {};
    bool need(int dev)
    {
        if (dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// *p = at least `bytes` of device memory owned by the context (slot is reused, contents
// are not preserved when it grows)
int lrb_ws_get(lrb_ctx *c, int slot, uint64_t bytes, void **p);

// (lrb_kernels.hip, for lrb_lists.hip) the reads the window lists leave out -- min_len bases and more --
// by the direct kernels: one atomic per window into the canonical half / one gather per window from the map
int lrb_k15_accum_half_long(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask, const uint64_t *d_code_off,
                            const uint64_t *d_mask_off, const uint32_t *d_lens, uint64_t n, uint32_t min_len,
                            uint32_t *d_half);
int lrb_cov_hist_map_long(lrb_ctx *c, const uint32_t *d_codes, const uint32_t *d_mask, const uint64_t *d_code_off,
                          const uint64_t *d_mask_off, const uint32_t *d_lens, uint64_t n, const uint8_t *d_map, int bins,
                          uint32_t *d_hist, uint32_t *d_sums);
// (lrb_lists.hip) reads per group of the window lists
uint64_t lrb_wl_group_reads(const lrb_ctx *c, uint64_t n, int bins, uint64_t total_bases);
bool lrb_wl_hist_fits(uint64_t reads_per_group, int bins);
int lrb_wl_reserve_scratch(lrb_ctx *c, uint64_t list_slots);

#endif
