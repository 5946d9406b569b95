// Device-side placement / exchange planner of the sharded filter (rbpf_plan.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace rbpf {

constexpr int kMaxWorld = 64;

struct PlanScalars {
  int start[kMaxWorld + 1];      // first sorted position of each ancestor rank's children
  int stay_cnt[kMaxWorld + 1];   // children that stay on their ancestor's rank
  int mv_off[kMaxWorld + 1];     // first moved-list index of each source rank
  int imp_start[kMaxWorld + 1];  // first moved-list index handed to each destination rank
  int M;                         // moved children in total
};

struct PlanBuffers {
  int *key, *counts, *offsets, *fill, *tmp, *order, *new_gid, *mv_child, *mv_src, *mv_q, *pref;
  int *slot_ids, *anc_bank, *send_idx;   // this rank's view
  PlanScalars* scalars;
  long long* counts_dev;                 // [4*world + 1]: my send counts, my recv counts, migrated, every rank's recv / send totals
};

// rec_off: first free record of the (persisting) receive buffer; imports get bank index nl + rec_off + position
hipError_t plan_run(const PlanBuffers& b, int N, int world, int nl, int me, const int* ai, const int* cur_gid,
                    int rec_off, hipStream_t s);

}  // namespace rbpf
