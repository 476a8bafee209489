// DeepSORT association stage: batched HIP kernels over all tracks / detections + host bookkeeping.
//
// Replaces the per-object torch/numpy/scipy code of the reference:
//   KalmanFilter.initiate/predict/project/update/gating_distance   deep_sort/sort/kalman_filter.py:54-256
//   NearestNeighborDistanceMetric.distance (_nn_cosine_distance)     deep_sort/sort/nn_matching.py:30-53,77-100,158-187
//   gate_cost_matrix + min_cost_matching thresholding                deep_sort/sort/linear_assignment.py:52,147-203
//   iou / iou_cost                                                   deep_sort/sort/iou_matching.py:5-91
//   scipy.optimize.linear_sum_assignment (third party)               call site linear_assignment.py:56
//   DeepSort.update output stage                                     deep_sort/deep_sort.py:63-88,108-114
// ALL track state lives in HBM: mean, covariance and the appearance gallery ring by slot id, and the integer lifecycle
// table (ids, hits, age, time_since_update, Tentative/Confirmed/Deleted, ring position, payload) in track-list order.
// The order-sensitive list bookkeeping (linear_assignment.py:58-72, tracker.py:56-93,115-176, track.py) runs in two
// single-workgroup kernels built from ordered compactions (ballot + prefix), so a frame is a fixed sequence of launches
// whose sizes are read from device memory: the host synchronises ONCE per frame - or once per batch of frames
// (step_batch, used by the pipeline) - to fetch the int32 rows.
#include "tracker_lsap_dev.h"

namespace yds {

// ============================================================================================ device-resident tracker
enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };
enum { METRIC_COSINE = 0, METRIC_EUCLIDEAN = 1 };

// Integer lifecycle table, one entry per track in track-list order (deep_sort/sort/track.py:63-79), in device memory.
struct TrackTable {
    int *slot, *id, *hits, *age, *tsu, *state, *n_feat, *head;
    float *payload;
};
constexpr int TAB_FIELDS = 9;

// device-side counters of one frame (ints)
enum Meta {
    M_T = 0,        // live tracks
    M_NEXT_ID,      // Tracker._next_id (tracker.py:47)
    M_NFREE,        // free gallery / Kalman slots
    M_TC, M_D,      // confirmed tracks, detections      (stage A problem: dims at &meta[M_TC])
    M_TB, M_DB,     // IOU-stage candidates, leftover detections (stage B problem: dims at &meta[M_TB])
    M_NA, M_NB,     // assigned pairs returned by the two LSAP solves
    M_NM_A,         // matches of stage A
    M_NUT_A,        // unmatched confirmed tracks of stage A
    M_NKEEP,        // of those, not eligible for the IOU stage
    M_NM,           // matches (both stages)
    M_NUT,          // unmatched tracks (final)
    M_NUD,          // unmatched detections (final) = new tracks
    M_NOUT,         // output rows
    M_MAXFEAT,      // largest gallery row count of any live track (sizes the nn_budget=None galleries)
    M_COUNT = 20
};

struct TrkDev {
    TrackTable tab, tmp;
    int *meta, *free_slots;
    float *mean, *cov, *gallery, *feats_n, *cost, *cost_b;      // cost: appearance stage [Tc][D], cost_b: IOU stage [Tb][Db]
    const float *tlwh, *payload;                    // this frame's detections
    int *conf_idx, *unconf_idx, *rows, *cols, *flag_r, *flag_c, *rej;
    int *matches, *um_t_a, *um_t_keep, *um_t, *um_d, *um_d2, *iou_cand;
    int *upd_slot, *upd_det, *upd_pos, *upd_row, *upd_of, *upd_of_tmp, *new_slot, *out_id;
    float *out_payload;
    int budget, unbounded, n_init, max_age;
    float max_dist, max_iou, flood_a, flood_b;      // thresholds of the two stages and the values their cost matrices are clamped to
    int euclid;
    // per-frame result block (device copy of what goes back to the host): header, rows, debug lists
    int *res;
    int res_out6, res_matches, res_um_t, res_um_d;   // int offsets inside res
};

// ordered compaction over [0, n) by a 256-thread workgroup: emit(i, rank) for every i with pred(i), ranks ascending
// with i; returns the number of hits (to every thread).  s_cnt: LDS int[4].
template <class Pred, class Emit> __device__ __forceinline__ int compact_ordered(int n, int *s_cnt, Pred pred, Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int total = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const bool p = i < n && pred(i);
        const unsigned long long b = __ballot(p);
        const int lane_rank = __popcll(b & ((1ull << lane) - 1ull)), wave_cnt = __popcll(b);
        if (lane == 0) s_cnt[wave] = wave_cnt;
        __syncthreads();
        int woff = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = s_cnt[w]; if (w < wave) woff += c; chunk += c; }
        if (p) emit(i, total + woff + lane_rank);
        __syncthreads();
        total += chunk;
    }
    return total;
}

// -------------------------------------------------------------------------------------------- one frame = three launches
// (round 4; rounds 1-3 ran ~14 launches per frame - begin, predict, cost, two LSAP forms, lists, IOU cost, two LSAP forms, lists,
//  Kalman, gallery, output - and under a saturated GPU every single-workgroup launch waits for a CU to drain: 280 us per frame at
//  30 persons inside the pipeline against 90 us on an idle chip)
//   trk_front_kernel   (T x ceil(D / 16) workgroups)  appearance cost + gate of the confirmed tracks, on a LOCAL Kalman prediction
//   trk_assoc_kernel   (one workgroup)                Tracker.predict for all tracks, both assignment problems with their list
//                                                     bookkeeping, the IOU cost in between, Tracker.update on the integer table,
//                                                     output selection, output rows of the tracks that were not updated
//   trk_back_kernel    (one workgroup per match / new track)  Kalman update / initiate, gallery row, output row of updated tracks

// deep_sort.py:73-87: state mean -> clipped corner box, int32 truncation, id, payload
__device__ __forceinline__ void write_out_row(const float *m, int id, float payload, int *o) {
    float w = m[2] * m[3], h = m[3];
    float x = m[0] - w / 2.f, y = m[1] - h / 2.f;
    float x2 = w + x, y2 = h + y;
    x = fmaxf(x, 0.f); y = fmaxf(y, 0.f);
    o[0] = (int)x; o[1] = (int)y; o[2] = (int)x2; o[3] = (int)y2; o[4] = id; o[5] = (int)payload;
}

// Appearance cost rows of the CONFIRMED tracks (tracker.py:56-78 gated_metric).  The state in memory is still the previous frame's
// posterior: the block predicts its own track in registers (the same kf_predict_body the association kernel applies in place right
// after this launch - same instruction sequence, same bits) and gates against that.  Row index = confirmed tracks in front of t in
// list order (the position conf_idx will give it).
__global__ __launch_bounds__(256) void trk_front_kernel(TrkDev d, int D) {
    __shared__ float pm[8], pP[64];
    const int t = blockIdx.x;
    if (t >= d.meta[M_T] || d.tab.state[t] != CONFIRMED) return;
    int row = 0;
    for (int base = 0; base < t; base += 256) {
        const int i = base + threadIdx.x;
        row += __syncthreads_count(i < t && d.tab.state[i] == CONFIRMED);
    }
    const int slot = d.tab.slot[t];
    if (threadIdx.x == 0) {
        float m[8], P[64];
        for (int i = 0; i < 8; ++i) m[i] = d.mean[(size_t)slot * 8 + i];
        for (int i = 0; i < 64; ++i) P[i] = d.cov[(size_t)slot * 64 + i];
        kf_predict_body(m, P);
        for (int i = 0; i < 8; ++i) pm[i] = m[i];
        for (int i = 0; i < 64; ++i) pP[i] = P[i];
    }
    __syncthreads();
    appearance_cost_block(d.gallery, slot, d.tab.n_feat[t], d.budget, d.feats_n, D, blockIdx.y * 16, pm, pP, d.tlwh, d.max_dist, d.flood_a, 1, d.euclid,
                          d.cost + (size_t)row * D);
}

// min_cost_matching list bookkeeping (linear_assignment.py:58-72) for one solved assignment problem:
//   unmatched detections = [columns not assigned, ascending] then [rejected pairs in row order], unmatched tracks likewise,
//   matches in row order; a pair is rejected when cost[row, col] > max_distance.
// row_name(r) / col_name(c) translate problem rows / columns to track indices / detection indices.
template <class RowName, class ColName>
__device__ __forceinline__ void assign_lists(const TrkDev &d, const float *cost, int *s_cnt, int nr, int nc, int n_pairs, float max_distance,
                                             RowName row_name, ColName col_name, int *matches, int &n_matches, int *um_t, int &n_um_t, int *um_d,
                                             int &n_um_d) {
    const int tid = threadIdx.x;
    for (int r = tid; r < nr; r += 256) d.flag_r[r] = 0;
    for (int c = tid; c < nc; c += 256) d.flag_c[c] = 0;
    __syncthreads();
    for (int k = tid; k < n_pairs; k += 256) {
        const int r = d.rows[k], c = d.cols[k];
        d.flag_r[r] = 1;
        d.flag_c[c] = 1;
        d.rej[k] = cost[(size_t)r * nc + c] > max_distance ? 1 : 0;
    }
    __syncthreads();
    const int d1 = compact_ordered(nc, s_cnt, [&](int c) { return d.flag_c[c] == 0; }, [&](int c, int q) { um_d[q] = col_name(c); });
    const int d2 = compact_ordered(n_pairs, s_cnt, [&](int k) { return d.rej[k] != 0; }, [&](int k, int q) { um_d[d1 + q] = col_name(d.cols[k]); });
    const int t1 = compact_ordered(nr, s_cnt, [&](int r) { return d.flag_r[r] == 0; }, [&](int r, int q) { um_t[q] = row_name(r); });
    const int t2 = compact_ordered(n_pairs, s_cnt, [&](int k) { return d.rej[k] != 0; }, [&](int k, int q) { um_t[t1 + q] = row_name(d.rows[k]); });
    const int nm = compact_ordered(n_pairs, s_cnt, [&](int k) { return d.rej[k] == 0; },
                                   [&](int k, int q) { matches[2 * (n_matches + q)] = row_name(d.rows[k]); matches[2 * (n_matches + q) + 1] = col_name(d.cols[k]); });
    n_um_d = d1 + d2;
    n_um_t = t1 + t2;
    n_matches += nm;
}

// The sequential part of a frame on ONE workgroup (every step consumes the lists of the one before; sizes live in registers):
//   Tracker.predict (track.py:110-123 + kalman_filter.py:89-124) for every track, confirmed / unconfirmed index lists (tracker.py:65-67)
//   stage A: linear assignment on the appearance cost (trk_front_kernel) + its lists, IOU-stage candidates (tracker.py:80-85)
//   stage B: IOU cost (iou_matching.py:46-91) + linear assignment + lists
//   Tracker.update on the integer table (tracker.py:129-176), deleted tracks dropped, output selection (deep_sort.py:67-71)
// smem_bytes of dynamic LDS belong to the assignment solvers (tracker_lsap_dev.h); lsap_state: global scratch for problems whose
// solver state exceeds the LDS (nullptr when the host-side bounds rule that out).
__global__ __launch_bounds__(256) void trk_assoc_kernel(TrkDev d, int D, char *lsap_state, int smem_bytes) {
    __shared__ int s_cnt[4];
    __shared__ int s_maxfeat;
    const int tid = threadIdx.x;
    int T = d.meta[M_T];
    for (int t = tid; t < T; t += 256) {
        d.tab.age[t]++; d.tab.tsu[t]++;
        const int slot = d.tab.slot[t];
        kf_predict_body(d.mean + (size_t)slot * 8, d.cov + (size_t)slot * 64);
    }
    __syncthreads();
    const int Tc = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] == CONFIRMED; }, [&](int t, int r) { d.conf_idx[r] = t; });
    const int Tu = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] != CONFIRMED; }, [&](int t, int r) { d.unconf_idx[r] = t; });
    __syncthreads();
    // ---- stage A (either side empty: nothing is solved, linear_assignment.py:48-49)
    lsap_solve_block(d.cost, Tc, D, d.rows, d.cols, d.meta + M_NA, lsap_state, smem_bytes);
    const int n_pairs_a = (Tc > 0 && D > 0) ? d.meta[M_NA] : 0;
    int n_matches = 0, n_um_t_a = 0, n_um_d_a = 0;
    assign_lists(d, d.cost, s_cnt, Tc, D, n_pairs_a, d.max_dist, [&](int r) { return d.conf_idx[r]; }, [&](int c) { return c; }, d.matches, n_matches,
                 d.um_t_a, n_um_t_a, d.um_d, n_um_d_a);
    __syncthreads();
    // IOU-stage candidates: unconfirmed tracks (index order) + unmatched confirmed tracks with time_since_update == 1
    for (int q = tid; q < Tu; q += 256) d.iou_cand[q] = d.unconf_idx[q];
    const int c1 = compact_ordered(n_um_t_a, s_cnt, [&](int q) { return d.tab.tsu[d.um_t_a[q]] == 1; }, [&](int q, int r) { d.iou_cand[Tu + r] = d.um_t_a[q]; });
    const int k_keep = compact_ordered(n_um_t_a, s_cnt, [&](int q) { return d.tab.tsu[d.um_t_a[q]] != 1; }, [&](int q, int r) { d.um_t_keep[r] = d.um_t_a[q]; });
    const int n_matches_a = n_matches;
    const int Tb = Tu + c1, Db = n_um_d_a;
    __syncthreads();
    // ---- stage B
    for (int idx = tid; idx < Tb * Db; idx += 256) {
        const int r = idx / Db, c = idx - r * Db, t = d.iou_cand[r];
        d.cost_b[idx] = iou_cost_entry(d.mean + (size_t)d.tab.slot[t] * 8, d.tlwh + (size_t)d.um_d[c] * 4, d.tab.tsu[t] > 1, d.max_iou, d.flood_b);
    }
    __syncthreads();
    lsap_solve_block(d.cost_b, Tb, Db, d.rows, d.cols, d.meta + M_NB, lsap_state, smem_bytes);
    const int n_pairs_b = (Tb > 0 && Db > 0) ? d.meta[M_NB] : 0;
    int n_um_t_b = 0, n_um_d = 0;
    int *um_t_b = d.um_t + k_keep;                                  // final unmatched tracks = [kept from stage A] + [stage B]
    assign_lists(d, d.cost_b, s_cnt, Tb, Db, n_pairs_b, d.max_iou, [&](int r) { return d.iou_cand[r]; }, [&](int c) { return d.um_d[c]; }, d.matches,
                 n_matches, um_t_b, n_um_t_b, d.um_d2, n_um_d);
    for (int q = tid; q < k_keep; q += 256) d.um_t[q] = d.um_t_keep[q];
    const int M = n_matches, n_um_t = k_keep + n_um_t_b, Nn = n_um_d;
    // entry k of the update list is handled by workgroup k of trk_back_kernel; upd_of[t] / upd_row[k] connect it with the output row
    for (int t = tid; t < T + Nn; t += 256) d.upd_of[t] = -1;
    for (int k = tid; k < M + Nn; k += 256) d.upd_row[k] = -1;
    __syncthreads();
    // ---- Track.update (track.py:125-144) for every match
    for (int k = tid; k < M; k += 256) {
        const int t = d.matches[2 * k], det = d.matches[2 * k + 1];
        int pos;
        if (d.unbounded) { pos = d.tab.n_feat[t]; d.tab.n_feat[t] = pos + 1; }
        else {
            pos = d.tab.head[t];
            d.tab.head[t] = (pos + 1) % d.budget;
            d.tab.n_feat[t] = min(d.tab.n_feat[t] + 1, d.budget);
        }
        const int hits = ++d.tab.hits[t];
        d.tab.tsu[t] = 0;
        if (d.tab.state[t] == TENTATIVE && hits >= d.n_init) d.tab.state[t] = CONFIRMED;
        d.tab.payload[t] = d.payload[det];
        d.upd_slot[k] = d.tab.slot[t]; d.upd_det[k] = det; d.upd_pos[k] = pos;
        d.upd_of[t] = k;
    }
    // ---- Track.mark_missed (track.py:146-152)
    for (int q = tid; q < n_um_t; q += 256) {
        const int t = d.um_t[q];
        if (d.tab.state[t] == TENTATIVE) d.tab.state[t] = DELETED;
        else if (d.tab.tsu[t] > d.max_age) d.tab.state[t] = DELETED;
    }
    // ---- Tracker._initiate_track (tracker.py:49-54) for the unmatched detections, in list order
    const int n_free = d.meta[M_NFREE], next_id = d.meta[M_NEXT_ID];
    for (int k = tid; k < Nn; k += 256) {
        const int t = T + k, slot = d.free_slots[n_free - 1 - k], det = d.um_d2[k];
        d.tab.slot[t] = slot; d.tab.id[t] = next_id + k; d.tab.hits[t] = 1; d.tab.age[t] = 1; d.tab.tsu[t] = 0; d.tab.state[t] = TENTATIVE;
        d.tab.n_feat[t] = 1; d.tab.head[t] = 1 % d.budget;
        d.tab.payload[t] = d.payload[det];
        d.new_slot[k] = slot;
        // the back kernel takes one combined list: entries [M, M + Nn) are the new tracks
        d.upd_slot[M + k] = slot; d.upd_det[M + k] = det; d.upd_pos[M + k] = 0;
    }
    __syncthreads();
    T += Nn;
    // ---- drop deleted tracks (tracker.py:162), slots go back to the free list
    const int alive = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] != DELETED; }, [&](int t, int r) {
        d.tmp.slot[r] = d.tab.slot[t]; d.tmp.id[r] = d.tab.id[t]; d.tmp.hits[r] = d.tab.hits[t]; d.tmp.age[r] = d.tab.age[t]; d.tmp.tsu[r] = d.tab.tsu[t];
        d.tmp.state[r] = d.tab.state[t]; d.tmp.n_feat[r] = d.tab.n_feat[t]; d.tmp.head[r] = d.tab.head[t]; d.tmp.payload[r] = d.tab.payload[t];
        d.upd_of_tmp[r] = d.upd_of[t];
    });
    const int freed = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] == DELETED; },
                                      [&](int t, int r) { d.free_slots[n_free - Nn + r] = d.tab.slot[t]; });
    __syncthreads();
    for (int t = tid; t < alive; t += 256) {
        d.tab.slot[t] = d.tmp.slot[t]; d.tab.id[t] = d.tmp.id[t]; d.tab.hits[t] = d.tmp.hits[t]; d.tab.age[t] = d.tmp.age[t]; d.tab.tsu[t] = d.tmp.tsu[t];
        d.tab.state[t] = d.tmp.state[t]; d.tab.n_feat[t] = d.tmp.n_feat[t]; d.tab.head[t] = d.tmp.head[t]; d.tab.payload[t] = d.tmp.payload[t];
        d.upd_of[t] = d.upd_of_tmp[t];
    }
    __syncthreads();
    // ---- output selection (deep_sort.py:67-71): confirmed and time_since_update <= 1, in list order.  A selected track that was
    //      updated in this frame gets its row from the workgroup that runs its Kalman update (trk_back_kernel); the others keep
    //      their predicted state, which this kernel wrote: their rows are final here.
    const int n_out = compact_ordered(alive, s_cnt, [&](int t) { return d.tab.state[t] == CONFIRMED && d.tab.tsu[t] <= 1; }, [&](int t, int r) {
        const int slot = d.tab.slot[t], id = d.tab.id[t], k = d.upd_of[t];
        const float payload = d.tab.payload[t];
        d.out_id[r] = id; d.out_payload[r] = payload;
        if (k >= 0) d.upd_row[k] = r;
        else write_out_row(d.mean + (size_t)slot * 8, id, payload, d.res + d.res_out6 + r * 6);
    });
    // ---- largest gallery of a live track: the host grows unbounded galleries by what is really held, not by frames seen
    if (tid == 0) s_maxfeat = 0;
    __syncthreads();
    {
        int mf = 0;
        for (int t = tid; t < alive; t += 256) mf = max(mf, d.tab.n_feat[t]);
        if (mf) atomicMax(&s_maxfeat, mf);
    }
    __syncthreads();
    // ---- debug lists for the parity tests
    for (int k = tid; k < 2 * M; k += 256) d.res[d.res_matches + k] = d.matches[k];
    for (int q = tid; q < n_um_t; q += 256) d.res[d.res_um_t + q] = d.um_t[q];
    for (int q = tid; q < Nn; q += 256) d.res[d.res_um_d + q] = d.um_d2[q];
    if (tid == 0) {
        d.meta[M_TC] = Tc; d.meta[M_D] = D; d.meta[M_TB] = Tb; d.meta[M_DB] = Db;
        d.meta[M_NM_A] = n_matches_a; d.meta[M_NUT_A] = n_um_t_a; d.meta[M_NKEEP] = k_keep;
        d.meta[M_T] = alive; d.meta[M_NEXT_ID] = next_id + Nn; d.meta[M_NFREE] = n_free - Nn + freed;
        d.meta[M_NM] = M; d.meta[M_NUT] = n_um_t; d.meta[M_NUD] = Nn; d.meta[M_NOUT] = n_out;
        d.meta[M_MAXFEAT] = s_maxfeat;
        for (int k = 0; k < M_COUNT; ++k) d.res[k] = d.meta[k];
    }
}

// Workgroup k: entry k of the update list - KalmanFilter.update for a match (kalman_filter.py:161-204 via tracker.py:143-150) or
// initiate for a new track (:54-87) on one thread, the gallery row (tracker.py:165-176 + nn_matching.py:152-155) on all of them,
// and the output row when the track is among the selected ones (deep_sort.py:73-87)
__global__ __launch_bounds__(128) void trk_back_kernel(TrkDev d) {
    const int k = blockIdx.x;
    const int M = d.meta[M_NM], Nn = d.meta[M_NUD];
    if (k >= M + Nn) return;
    const int slot = d.upd_slot[k], det = d.upd_det[k];
    if (threadIdx.x == 0) {
        const float *b = d.tlwh + (size_t)det * 4;
        float *m = d.mean + (size_t)slot * 8, *P = d.cov + (size_t)slot * 64;
        if (k < M) {
            float z[4];
            to_xyah(b, z);
            kf_update_body(m, P, z);
        } else {
            kf_initiate_body(m, P, b);
        }
        const int r = d.upd_row[k];
        if (r >= 0) write_out_row(m, d.out_id[r], d.out_payload[r], d.res + d.res_out6 + r * 6);
    }
    float *dst = d.gallery + ((size_t)slot * d.budget + d.upd_pos[k]) * EMB;
    const float *src = d.feats_n + (size_t)det * EMB;
    for (int c = threadIdx.x; c < EMB; c += 128) dst[c] = src[c];
}

// ============================================================================================ host
class Tracker : public TrackerIface {
public:
    // budget <= 0: nn_budget=None, every track keeps all its features (nn_matching.py:152-154) - the per-track row
    // capacity `budget` then doubles whenever a gallery could fill up; metric: cosine | euclidean (nn_matching.py:128-134)
    Tracker(double max_dist, double max_iou, int max_age, int n_init, int budget, int metric = METRIC_COSINE)
        : max_dist(max_dist), max_iou(max_iou), max_age(max_age), n_init(n_init), budget(budget > 0 ? budget : 32), unbounded(budget <= 0),
          metric(metric) {
        if (metric != METRIC_COSINE && metric != METRIC_EUCLIDEAN) fail("Invalid metric; must be either 'euclidean' or 'cosine'");
        stream = make_stream(true);
        feats_stage.st = feats_n.st = cost_dev.st = &stream;
        det_lists.st = &stream;
        meta.alloc(M_COUNT);
        int init[M_COUNT] = {};
        init[M_NEXT_ID] = 1;
        YDS_HIP(hipMemcpy(meta.p, init, sizeof init, hipMemcpyHostToDevice));
        grow(256);
    }
    ~Tracker() override {
        if (res_host) (void)hipHostFree(res_host);
        if (in_host) (void)hipHostFree(in_host);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int num_tracks() const override { return T_host; }

    static TrackTable table_at(int *base, int cap) {
        TrackTable t;
        t.slot = base; t.id = base + cap; t.hits = base + 2 * cap; t.age = base + 3 * cap; t.tsu = base + 4 * cap; t.state = base + 5 * cap;
        t.n_feat = base + 6 * cap; t.head = base + 7 * cap; t.payload = reinterpret_cast<float *>(base + 8 * cap);
        return t;
    }

    // capacity = number of slots = maximum number of live tracks.  Nothing may be in flight on the stream.
    void grow(int cap) {
        YDS_HIP(hipStreamSynchronize(stream));
        DevBuf<float> m((size_t)cap * 8), c((size_t)cap * 64), g((size_t)cap * budget * EMB);
        DevBuf<int> tab((size_t)cap * TAB_FIELDS), tmp((size_t)cap * TAB_FIELDS), fs(cap), lists((size_t)cap * 13);
        if (capacity) {
            YDS_HIP(hipMemcpy(m.p, mean.p, (size_t)capacity * 8 * 4, hipMemcpyDeviceToDevice));
            YDS_HIP(hipMemcpy(c.p, cov.p, (size_t)capacity * 64 * 4, hipMemcpyDeviceToDevice));
            YDS_HIP(hipMemcpy(g.p, gallery.p, (size_t)capacity * budget * EMB * 4, hipMemcpyDeviceToDevice));
            for (int f = 0; f < TAB_FIELDS; ++f)
                YDS_HIP(hipMemcpy(tab.p + (size_t)f * cap, table.p + (size_t)f * capacity, (size_t)capacity * 4, hipMemcpyDeviceToDevice));
        }
        // free list: existing entries, then the new slots on top (popped first); the count lives in meta[M_NFREE]
        int n_free = 0;
        YDS_HIP(hipMemcpy(&n_free, meta.p + M_NFREE, 4, hipMemcpyDeviceToHost));
        std::vector<int> fl(cap);
        if (capacity && n_free) YDS_HIP(hipMemcpy(fl.data(), free_slots.p, (size_t)n_free * 4, hipMemcpyDeviceToHost));
        for (int s = cap - 1; s >= capacity; --s) fl[n_free++] = s;
        YDS_HIP(hipMemcpy(fs.p, fl.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
        YDS_HIP(hipMemcpy(meta.p + M_NFREE, &n_free, 4, hipMemcpyHostToDevice));
        mean = std::move(m); cov = std::move(c); gallery = std::move(g); table = std::move(tab); table_tmp = std::move(tmp);
        free_slots = std::move(fs); track_lists = std::move(lists);
        capacity = cap;
    }
    // nn_budget=None: double the per-track row capacity, keeping every slot's rows
    void grow_budget() {
        YDS_HIP(hipStreamSynchronize(stream));
        const int nb = budget * 2;
        size_t free_b = 0, total_b = 0;
        YDS_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t need = (size_t)capacity * nb * EMB * 4;
        if (need > free_b)
            fail("tracker: nn_budget=None galleries would need %zu MB for %d slots x %d rows (%zu MB free); use a finite nn_budget",
                 need >> 20, capacity, nb, free_b >> 20);
        DevBuf<float> g((size_t)capacity * nb * EMB);
        YDS_HIP(hipMemcpy2D(g.p, (size_t)nb * EMB * 4, gallery.p, (size_t)budget * EMB * 4, (size_t)budget * EMB * 4, capacity, hipMemcpyDeviceToDevice));
        gallery = std::move(g);
        budget = nb;
    }

    struct FrameIn { const float *tlwh; const float *feats; bool feats_on_device; const int *feat_rows; const float *payload; int D; };

    // Enqueues one frame (three launches, see the kernels); T_ub = host-side upper bound of the live track count when it starts.
    // feats_n_frame: this frame's normalised embeddings when the caller has normalised the whole batch in one launch, else nullptr.
    // Returns the int offset of this frame's result block inside res_host (pinned host memory the kernels store into: no copy
    // command at the end).
    size_t enqueue(const FrameIn &f, int T_ub, size_t in_off, size_t res_off, size_t *res_len, int *out_cap, float *feats_n_frame) {
        const int D = f.D, Dn = std::max(D, 1), Tn = std::max(T_ub, 1);
        // ---- inputs: tlwh, payload (and feat_rows) were packed into in_host by the caller; one H2D per batch
        TrkDev d;
        d.tab = table_at(table.p, capacity); d.tmp = table_at(table_tmp.p, capacity);
        d.meta = meta.p; d.free_slots = free_slots.p;
        d.mean = mean.p; d.cov = cov.p; d.gallery = gallery.p;
        if (!feats_n_frame) feats_n.ensure_keep((size_t)Dn * EMB);
        const size_t cost_n = (size_t)(Tn + Dn) * Dn;
        cost_dev.ensure_keep(2 * cost_n);
        det_lists.ensure_keep((size_t)Dn * 8 + (size_t)(Tn + Dn) * 8);
        d.feats_n = feats_n_frame ? feats_n_frame : feats_n.p;
        d.cost = cost_dev.p; d.cost_b = cost_dev.p + cost_n;
        d.tlwh = reinterpret_cast<const float *>(in_dev.p + in_off);
        d.payload = d.tlwh + (size_t)D * 4;
        const int *feat_rows_dev = f.feat_rows ? reinterpret_cast<const int *>(d.payload + D) : nullptr;
        int *tl = track_lists.p;                                  // 13 lists of `capacity` ints
        const int cap = capacity;
        d.conf_idx = tl; d.unconf_idx = tl + cap; d.flag_r = tl + 2 * cap; d.um_t_a = tl + 3 * cap; d.um_t_keep = tl + 4 * cap; d.um_t = tl + 5 * cap;
        d.iou_cand = tl + 6 * cap; d.upd_of = tl + 7 * cap; d.out_id = tl + 8 * cap; d.out_payload = reinterpret_cast<float *>(tl + 9 * cap);
        d.rows = tl + 10 * cap; d.cols = tl + 11 * cap; d.upd_of_tmp = tl + 12 * cap;
        int *dl = det_lists.p;                                    // per-detection lists (D) and per-(track+det) lists
        d.flag_c = dl; d.um_d = dl + Dn; d.um_d2 = dl + 2 * Dn; d.new_slot = dl + 3 * Dn; d.rej = dl + 4 * Dn;
        int *pl = dl + 8 * (size_t)Dn;
        const int P = Tn + Dn;
        d.matches = pl; d.upd_slot = pl + 2 * P; d.upd_det = pl + 3 * P; d.upd_pos = pl + 4 * P; d.upd_row = pl + 5 * P;
        d.budget = budget; d.unbounded = unbounded ? 1 : 0; d.n_init = n_init; d.max_age = max_age;
        d.max_dist = (float)max_dist; d.max_iou = (float)max_iou;
        d.flood_a = (float)(max_dist + 1e-5); d.flood_b = (float)(max_iou + 1e-5);           // linear_assignment.py:52
        d.euclid = metric == METRIC_EUCLIDEAN ? 1 : 0;
        // ---- result block layout: header | out6 rows | matches | unmatched tracks | unmatched detections
        const int rows_cap = T_ub + D, mcap = T_ub + D;
        d.res_out6 = M_COUNT; d.res_matches = d.res_out6 + rows_cap * 6; d.res_um_t = d.res_matches + 2 * mcap; d.res_um_d = d.res_um_t + T_ub + D;
        *res_len = (size_t)d.res_um_d + D + 1;
        *out_cap = rows_cap;
        d.res = res_host + res_off;

        // ---- kernels (sizes come from device memory; the grids use the host-side upper bounds)
        if (!feats_n_frame && D) {
            const float *feats_dev = f.feats;
            if (!f.feats_on_device) {
                int n_rows = D;
                if (f.feat_rows) for (int k = 0; k < D; ++k) n_rows = std::max(n_rows, f.feat_rows[k] + 1);
                feats_stage.ensure_keep((size_t)n_rows * EMB);
                YDS_HIP(hipMemcpyAsync(feats_stage.p, f.feats, (size_t)n_rows * EMB * 4, hipMemcpyHostToDevice, stream));
                feats_dev = feats_stage.p;
            }
            hipLaunchKernelGGL(normalize_rows_kernel, dim3((D + 3) / 4), dim3(256), 0, stream, feats_dev, feat_rows_dev, feats_n.p, D,
                               metric == METRIC_COSINE ? 1 : 0);     // x / ||x|| once per frame (nn_matching.py:50-52); euclidean: as is
        }
        if (T_ub && D) hipLaunchKernelGGL(trk_front_kernel, dim3(T_ub, (D + 15) / 16), dim3(256), 0, stream, d, D);
        // the assignment solvers get the whole LDS; their state moves to a global scratch when even that is too small (~3000 rows)
        static bool attr_set = false;
        if (!attr_set) {
            YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(trk_assoc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSAP_LDS_MAX));
            attr_set = true;
        }
        const size_t lsap_state = (size_t)std::max(Tn, Dn) * LSAP_STATE_BYTES;
        if (lsap_state > LSAP_LDS_MAX && lsap_scratch.n < lsap_state) {
            YDS_HIP(hipStreamSynchronize(stream));                   // nothing may still use the old scratch
            lsap_scratch.alloc(lsap_state);
        }
        hipLaunchKernelGGL(trk_assoc_kernel, dim3(1), dim3(256), LSAP_LDS_MAX, stream, d, D, lsap_state > LSAP_LDS_MAX ? lsap_scratch.p : (char *)nullptr,
                           (int)LSAP_LDS_MAX);
        if (D) hipLaunchKernelGGL(trk_back_kernel, dim3(D), dim3(128), 0, stream, d);
        YDS_HIP(hipGetLastError());
        return res_off;
    }

    // One or several frames, in order, with ONE host synchronisation at the end.  counts[b] = rows of frame b.
    void run(const FrameIn *frames, int n_frames, int32_t *const *out6, const int *caps, int *counts) {
        // ---- capacity for the worst case: every detection of the batch starts a track, every frame adds a gallery row
        int D_sum = 0;
        for (int b = 0; b < n_frames; ++b) D_sum += frames[b].D;
        if (T_host + D_sum > capacity) { int c = capacity; while (c < T_host + D_sum) c *= 2; grow(c); }
        // (max_rows = the largest gallery any LIVE track held after the last synchronised frame, from the result header;
        //  every frame of this call can add one row to it)
        if (unbounded) while (max_rows + n_frames + 1 > budget) grow_budget();
        // ---- inputs of all frames in one pinned block, one upload
        std::vector<size_t> in_off(n_frames);
        size_t in_total = 0;
        for (int b = 0; b < n_frames; ++b) { in_off[b] = in_total; in_total += (size_t)frames[b].D * 6 + 4; }
        if (in_total > in_cap) {
            if (in_host) (void)hipHostFree(in_host);
            in_cap = in_total * 2;
            YDS_HIP(hipHostMalloc((void **)&in_host, in_cap * sizeof(int)));
            in_dev.alloc(in_cap);
        }
        for (int b = 0; b < n_frames; ++b) {
            const FrameIn &f = frames[b];
            float *dst = reinterpret_cast<float *>(in_host + in_off[b]);
            if (f.D) {
                memcpy(dst, f.tlwh, (size_t)f.D * 16);
                memcpy(dst + (size_t)f.D * 4, f.payload, (size_t)f.D * 4);
                if (f.feat_rows) memcpy(dst + (size_t)f.D * 5, f.feat_rows, (size_t)f.D * 4);
            }
        }
        YDS_HIP(hipMemcpyAsync(in_dev.p, in_host, in_total * sizeof(int), hipMemcpyHostToDevice, stream));
        // ---- result blocks
        std::vector<size_t> res_off(n_frames), res_len(n_frames);
        std::vector<int> out_cap(n_frames);
        size_t res_total = 0;
        {
            int T_ub = T_host;
            for (int b = 0; b < n_frames; ++b) { res_off[b] = res_total; res_total += (size_t)M_COUNT + (size_t)(T_ub + frames[b].D) * 10 + frames[b].D + 8; T_ub += frames[b].D; }
        }
        if (res_total > res_cap) {
            YDS_HIP(hipStreamSynchronize(stream));
            if (res_host) (void)hipHostFree(res_host);
            res_cap = res_total * 2;
            YDS_HIP(hipHostMalloc((void **)&res_host, res_cap * sizeof(int)));
        }
        // ---- embeddings of the whole batch normalised in ONE launch when they already sit back to back on the device (the
        //      pipeline's ReID pass writes them that way): x / ||x|| once per frame in the reference (nn_matching.py:50-52)
        bool batch_norm = n_frames > 1 && D_sum > 0;
        {
            const float *expect = nullptr;
            for (int b = 0; b < n_frames && batch_norm; ++b) {
                const FrameIn &f = frames[b];
                if (!f.D) continue;
                if (!f.feats_on_device || f.feat_rows || (expect && f.feats != expect)) batch_norm = false;
                expect = f.feats + (size_t)f.D * EMB;
            }
        }
        std::vector<float *> fn(n_frames, nullptr);
        if (batch_norm) {
            feats_n.ensure_keep((size_t)D_sum * EMB);
            const float *first_feats = nullptr;
            size_t off = 0;
            for (int b = 0; b < n_frames; ++b) {
                if (frames[b].D && !first_feats) first_feats = frames[b].feats;
                fn[b] = feats_n.p + off * EMB;
                off += frames[b].D;
            }
            hipLaunchKernelGGL(normalize_rows_kernel, dim3((D_sum + 3) / 4), dim3(256), 0, stream, first_feats, (const int *)nullptr, feats_n.p, D_sum,
                               metric == METRIC_COSINE ? 1 : 0);
        }
        int T_ub = T_host;
        for (int b = 0; b < n_frames; ++b) {
            enqueue(frames[b], T_ub, in_off[b], res_off[b], &res_len[b], &out_cap[b], fn[b]);
            T_ub += frames[b].D;
        }
        // (the result blocks are in host memory already: a device-to-host copy command would queue behind frame uploads)
        YDS_HIP(hipStreamSynchronize(stream));
        for (int b = 0; b < n_frames; ++b) {
            const int *r = res_host + res_off[b];
            const int m = r[M_NOUT];
            if (m > caps[b]) fail("tracker: %d output rows exceed the caller's capacity %d", m, caps[b]);
            if (m) memcpy(out6[b], r + M_COUNT, (size_t)m * 6 * sizeof(int));
            counts[b] = m;
        }
        // ---- host-side mirror of the last frame (debug lists for the parity tests, live track count)
        {
            const int b = n_frames - 1, D = frames[b].D;
            int T_before = T_host;
            for (int k = 0; k < b; ++k) T_before += frames[k].D;         // upper bound used for that frame's layout
            const int *r = res_host + res_off[b];
            const int rows_cap = T_before + D;
            const int *pm = r + M_COUNT + rows_cap * 6, *pt = pm + 2 * rows_cap, *pd = pt + rows_cap;
            last_matches.assign(r[M_NM], {0, 0});
            for (int k = 0; k < r[M_NM]; ++k) last_matches[k] = {pm[2 * k], pm[2 * k + 1]};
            last_um_t.assign(pt, pt + r[M_NUT]);
            std::sort(last_um_t.begin(), last_um_t.end());
            last_um_d.assign(pd, pd + r[M_NUD]);
            T_host = r[M_T];
            max_rows = r[M_MAXFEAT];
        }
    }

    int step(const float *tlwh_host, const float *feats, bool feats_on_device, const float *payload, int D, int32_t *out6, int cap) override {
        return step_sel(tlwh_host, feats, feats_on_device, nullptr, payload, D, out6, cap);
    }
    // feat_rows (optional): detection d uses row feat_rows[d] of `feats` (tracker-side NMS keeps a subset in pick order)
    int step_sel(const float *tlwh_host, const float *feats, bool feats_on_device, const int *feat_rows, const float *payload, int D,
                 int32_t *out6, int cap) {
        FrameIn f{tlwh_host, feats, feats_on_device, feat_rows, payload, D};
        int count = 0;
        run(&f, 1, &out6, &cap, &count);
        return count;
    }
    // frames of one batch, in order, one synchronisation (the pipeline's association stage); skip[b]: tracker not called
    void wait_for(hipEvent_t ev) override { YDS_HIP(hipStreamWaitEvent(stream, ev, 0)); }
    void step_batch(int n, const float *tlwh_host, const int *first, const float *feats_dev, const float *payload, const char *skip, int32_t *out6,
                    int cap, int32_t *counts) override {
        std::vector<FrameIn> fr;
        std::vector<int32_t *> outs;
        std::vector<int> caps, cnt, which;
        for (int b = 0; b < n; ++b) {
            if (skip && skip[b]) { counts[b] = -1; continue; }
            const int D = first[b + 1] - first[b];
            fr.push_back(FrameIn{tlwh_host + (size_t)first[b] * 4, feats_dev + (size_t)first[b] * EMB, true, nullptr, payload + first[b], D});
            outs.push_back(out6 + (size_t)b * cap * 6);
            caps.push_back(cap);
            which.push_back(b);
        }
        cnt.assign(fr.size(), 0);
        if (!fr.empty()) run(fr.data(), (int)fr.size(), outs.data(), caps.data(), cnt.data());
        for (size_t k = 0; k < which.size(); ++k) counts[which[k]] = cnt[k];
    }

    // host copy of the integer table (parity tests, DeepSort.tracker.tracks)
    struct HostTable { std::vector<int> slot, id, hits, age, tsu, state, n_feat; };
    HostTable read_table() {
        HostTable h;
        const int T = T_host;
        auto get = [&](int field, std::vector<int> &v) {
            v.resize(T);
            if (T) YDS_HIP(hipMemcpy(v.data(), table.p + (size_t)field * capacity, (size_t)T * 4, hipMemcpyDeviceToHost));
        };
        get(0, h.slot); get(1, h.id); get(2, h.hits); get(3, h.age); get(4, h.tsu); get(5, h.state); get(6, h.n_feat);
        return h;
    }

    double max_dist, max_iou;       // python floats in the reference; fp32 roundings are taken where torch/numpy take them
    int max_age, n_init, budget;
    bool unbounded = false;
    int metric = METRIC_COSINE;
    int capacity = 0;
    int T_host = 0;                 // live tracks after the last synchronised frame
    int max_rows = 1;               // largest gallery row count of a live track after the last synchronised frame
    template <class T> struct GrowBuf : DevBuf<T> {
        // ensure() that never shrinks and - unlike DevBuf::ensure - may only be called while nothing that uses the old buffer is
        // in flight; growth is rare (sizes follow the largest frame seen), so it simply drains the stream first
        hipStream_t *st = nullptr;
        void ensure_keep(size_t count) {
            if (count <= this->n) return;
            if (st && *st) (void)hipStreamSynchronize(*st);
            this->alloc(count + count / 2);
        }
    };
    DevBuf<float> mean, cov, gallery /* cosine: rows stored normalised */;
    DevBuf<int> table, table_tmp, free_slots, track_lists, meta, in_dev;
    GrowBuf<float> feats_stage, feats_n, cost_dev;
    GrowBuf<int> det_lists;
    DevBuf<char> lsap_scratch;
    int *res_host = nullptr, *in_host = nullptr;
    size_t res_cap = 0, in_cap = 0;
    std::vector<std::pair<int, int>> last_matches;
    std::vector<int> last_um_t, last_um_d;
    hipStream_t stream = nullptr;
};

}  // namespace yds

// ============================================================================================ C ABI
static inline yds::Tracker *impl(yds_trk *h) { return static_cast<yds::Tracker *>(h->t); }
static inline const yds::Tracker *impl(const yds_trk *h) { return static_cast<const yds::Tracker *>(h->t); }

namespace {
struct Scratch {
    hipStream_t s = nullptr;
    ~Scratch() { if (s) (void)hipStreamDestroy(s); }
    hipStream_t stream() { if (!s) YDS_HIP(hipStreamCreate(&s)); return s; }
};
Scratch g_scratch;
std::vector<int> iota(int n) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = i; return v; }
}  // namespace

extern "C" {

yds_trk *yds_tracker_create(double max_dist, double max_iou_distance, int max_age, int n_init, int nn_budget) {
    YDS_API_BEGIN
    return new yds_trk{new yds::Tracker(max_dist, max_iou_distance, max_age, n_init, nn_budget)};
    YDS_API_END_PTR
}
yds_trk *yds_tracker_create_ex(double max_dist, double max_iou_distance, int max_age, int n_init, int nn_budget, int metric) {
    YDS_API_BEGIN
    return new yds_trk{new yds::Tracker(max_dist, max_iou_distance, max_age, n_init, nn_budget, metric)};
    YDS_API_END_PTR
}
int yds_tracker_step_sel(yds_trk *t, const float *tlwh, const float *feats, int feats_on_device, const int32_t *feat_rows, const float *payload,
                         int D, int32_t *out6, int cap, int *m_out, int32_t *dbg_matches, int dbg_cap, int *n_matches) {
    YDS_API_BEGIN
    *m_out = impl(t)->step_sel(tlwh, feats, feats_on_device != 0, feat_rows, payload, D, out6, cap);
    if (n_matches) {
        const auto &lm = impl(t)->last_matches;
        *n_matches = (int)lm.size();
        if (dbg_matches) {
            if ((int)lm.size() > dbg_cap) yds::fail("tracker: %zu matches exceed dbg_cap %d", lm.size(), dbg_cap);
            for (size_t k = 0; k < lm.size(); ++k) { dbg_matches[2 * k] = lm[k].first; dbg_matches[2 * k + 1] = lm[k].second; }
        }
    }
    YDS_API_END
}
void yds_tracker_destroy(yds_trk *t) {
    if (t) { delete t->t; delete t; }
}
int yds_tracker_step(yds_trk *t, const float *tlwh, const float *feats, const float *payload, int D, int32_t *out6, int cap, int *m_out,
                     int32_t *dbg_matches, int dbg_cap, int *n_matches) {
    YDS_API_BEGIN
    *m_out = impl(t)->step(tlwh, feats, false, payload, D, out6, cap);
    if (n_matches) {
        const auto &lm = impl(t)->last_matches;
        *n_matches = (int)lm.size();
        if (dbg_matches) {
            if ((int)lm.size() > dbg_cap) yds::fail("tracker: %zu matches exceed dbg_cap %d", lm.size(), dbg_cap);
            for (size_t k = 0; k < lm.size(); ++k) { dbg_matches[2 * k] = lm[k].first; dbg_matches[2 * k + 1] = lm[k].second; }
        }
    }
    YDS_API_END
}
int yds_tracker_step_dev(yds_trk *t, const float *tlwh, const float *feats_dev, const float *payload, int D, int32_t *out6, int cap, int *m_out) {
    YDS_API_BEGIN
    *m_out = impl(t)->step(tlwh, feats_dev, true, payload, D, out6, cap);
    YDS_API_END
}
int yds_tracker_num_tracks(const yds_trk *t) { return t->t->num_tracks(); }
int yds_tracker_get_state(yds_trk *t, int32_t *ids, int32_t *state, int32_t *tsu, int32_t *hits, float *mean8, float *cov64, int cap, int *T) {
    YDS_API_BEGIN
    yds::Tracker *k = impl(t);
    const int n = k->T_host;
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    *T = n;
    const yds::Tracker::HostTable h = k->read_table();          // the table lives on the device; this is a debug / parity read-back
    std::vector<float> m, c;
    if (mean8 && n) { m.resize((size_t)k->capacity * 8); YDS_HIP(hipMemcpy(m.data(), k->mean.p, m.size() * 4, hipMemcpyDeviceToHost)); }
    if (cov64 && n) { c.resize((size_t)k->capacity * 64); YDS_HIP(hipMemcpy(c.data(), k->cov.p, c.size() * 4, hipMemcpyDeviceToHost)); }
    for (int i = 0; i < n; ++i) {
        if (ids) ids[i] = h.id[i];
        if (state) state[i] = h.state[i];
        if (tsu) tsu[i] = h.tsu[i];
        if (hits) hits[i] = h.hits[i];
        if (mean8) memcpy(mean8 + (size_t)i * 8, &m[(size_t)h.slot[i] * 8], 32);
        if (cov64) memcpy(cov64 + (size_t)i * 64, &c[(size_t)h.slot[i] * 64], 256);
    }
    YDS_API_END
}
int yds_tracker_gallery_rows(const yds_trk *t) { return impl(t)->budget; }
int yds_tracker_get_payload(yds_trk *t, float *payload, int cap) {
    YDS_API_BEGIN
    yds::Tracker *k = impl(t);
    const int n = k->T_host;
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    if (n) YDS_HIP(hipMemcpy(payload, k->table.p + (size_t)8 * k->capacity, (size_t)n * 4, hipMemcpyDeviceToHost));
    YDS_API_END
}
int yds_tracker_get_age(yds_trk *t, int32_t *age, int cap) {
    YDS_API_BEGIN
    yds::Tracker *k = impl(t);
    const int n = k->T_host;
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    if (n) YDS_HIP(hipMemcpy(age, k->table.p + (size_t)3 * k->capacity, (size_t)n * 4, hipMemcpyDeviceToHost));
    YDS_API_END
}
int yds_tracker_last_unmatched(yds_trk *t, int32_t *um_tracks, int cap_t, int *n_t, int32_t *um_dets, int cap_d, int *n_d) {
    YDS_API_BEGIN
    const auto &a = impl(t)->last_um_t, &b = impl(t)->last_um_d;
    if ((int)a.size() > cap_t || (int)b.size() > cap_d) yds::fail("tracker: unmatched lists exceed caller capacity");
    *n_t = (int)a.size(); *n_d = (int)b.size();
    for (size_t i = 0; i < a.size(); ++i) um_tracks[i] = a[i];
    for (size_t i = 0; i < b.size(); ++i) um_dets[i] = b[i];
    YDS_API_END
}

// ---- stand-alone primitives for parity tests ------------------------------------------------------
int yds_lsap(const float *cost_host, int nr, int nc, int32_t *rows, int32_t *cols, int *n_out) {
    YDS_API_BEGIN
    using namespace yds;
    int n = std::min(nr, nc);
    *n_out = n;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> c; c.upload(cost_host, (size_t)nr * nc, s);
    DevBuf<int> r(n), cc(n);
    DevBuf<char> scratch;
    launch_lsap(c.p, nr, nc, nullptr, r.p, cc.p, nullptr, scratch, s);
    YDS_HIP(hipMemcpyAsync(rows, r.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cols, cc.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_lsap_bench(const float *cost_host, int nr, int nc, int iters, double *avg_us) {
    YDS_API_BEGIN
    using namespace yds;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> c; c.upload(cost_host, (size_t)nr * nc, s);
    const int n = std::max(std::min(nr, nc), 1);
    DevBuf<int> r(n), cc(n);
    DevBuf<char> scratch;
    hipEvent_t e0, e1;
    YDS_HIP(hipEventCreate(&e0)); YDS_HIP(hipEventCreate(&e1));
    launch_lsap(c.p, nr, nc, nullptr, r.p, cc.p, nullptr, scratch, s);
    YDS_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_lsap(c.p, nr, nc, nullptr, r.p, cc.p, nullptr, scratch, s);
    YDS_HIP(hipEventRecord(e1, s));
    YDS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    YDS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1e3 / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    YDS_API_END
}
int yds_kalman_predict(float *mean_host, float *cov_host, int T) {
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c; DevBuf<int> sl;
    m.upload(mean_host, (size_t)T * 8, s); c.upload(cov_host, (size_t)T * 64, s);
    auto v = iota(T); sl.upload(v.data(), T, s);
    hipLaunchKernelGGL(kf_predict_kernel, dim3((T + 63) / 64), dim3(64), 0, s, m.p, c.p, sl.p, T);
    YDS_HIP(hipMemcpyAsync(mean_host, m.p, (size_t)T * 32, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov_host, c.p, (size_t)T * 256, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_update(float *mean_host, float *cov_host, const float *xyah_host, int M) {
    YDS_API_BEGIN
    using namespace yds;
    if (M == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, z; DevBuf<int> sl;
    m.upload(mean_host, (size_t)M * 8, s); c.upload(cov_host, (size_t)M * 64, s); z.upload(xyah_host, (size_t)M * 4, s);
    auto v = iota(M); sl.upload(v.data(), M, s);
    hipLaunchKernelGGL(kf_update_kernel, dim3((M + 63) / 64), dim3(64), 0, s, m.p, c.p, sl.p, z.p, M);
    YDS_HIP(hipMemcpyAsync(mean_host, m.p, (size_t)M * 32, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov_host, c.p, (size_t)M * 256, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_gating(const float *mean_host, const float *cov_host, int T, const float *xyah_host, int D, float *out) {
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, z, o((size_t)T * D); DevBuf<int> sl;
    m.upload(mean_host, (size_t)T * 8, s); c.upload(cov_host, (size_t)T * 64, s); z.upload(xyah_host, (size_t)D * 4, s);
    auto v = iota(T); sl.upload(v.data(), T, s);
    hipLaunchKernelGGL(gating_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, c.p, sl.p, T, z.p, D, o.p);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_gating_ex(const float *mean_host, const float *cov_host, int T, const float *xyah_host, int D, int only_position, float *out) {
    if (only_position) return yds_kalman_gating(mean_host, cov_host, T, xyah_host, D, out);
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, z, o((size_t)T * D);
    m.upload(mean_host, (size_t)T * 8, s); c.upload(cov_host, (size_t)T * 64, s); z.upload(xyah_host, (size_t)D * 4, s);
    hipLaunchKernelGGL(gating4_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, c.p, T, z.p, D, o.p);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_initiate(const float *xyah_host, int n, float *mean_host, float *cov_host) {
    YDS_API_BEGIN
    using namespace yds;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> z, m((size_t)n * 8), c((size_t)n * 64);
    z.upload(xyah_host, (size_t)n * 4, s);
    hipLaunchKernelGGL(kf_initiate_xyah_kernel, dim3((n + 63) / 64), dim3(64), 0, s, z.p, m.p, c.p, n);
    YDS_HIP(hipMemcpyAsync(mean_host, m.p, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov_host, c.p, (size_t)n * 256, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_project(const float *mean_host, const float *cov_host, int n, float *mean4_host, float *cov16_host) {
    YDS_API_BEGIN
    using namespace yds;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, m4((size_t)n * 4), c16((size_t)n * 16);
    m.upload(mean_host, (size_t)n * 8, s); c.upload(cov_host, (size_t)n * 64, s);
    hipLaunchKernelGGL(kf_project_kernel, dim3((n + 63) / 64), dim3(64), 0, s, m.p, c.p, m4.p, c16.p, n);
    YDS_HIP(hipMemcpyAsync(mean4_host, m4.p, (size_t)n * 16, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov16_host, c16.p, (size_t)n * 64, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_iou_cost(const float *track_tlwh_host, int T, const float *det_tlwh_host, int D, float *out) {
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    // tracks are given as tlwh: rebuild the (x, y, a, h) means the kernel reads
    std::vector<float> mean((size_t)T * 8, 0.f);
    for (int t = 0; t < T; ++t) {
        const float *b = track_tlwh_host + t * 4;
        mean[t * 8 + 0] = b[0] + b[2] / 2.f; mean[t * 8 + 1] = b[1] + b[3] / 2.f; mean[t * 8 + 2] = b[2] / b[3]; mean[t * 8 + 3] = b[3];
    }
    DevBuf<float> m, dt, o((size_t)T * D); DevBuf<int> sl, di;
    m.upload(mean.data(), mean.size(), s); dt.upload(det_tlwh_host, (size_t)D * 4, s);
    auto v = iota(T); sl.upload(v.data(), T, s);
    auto w = iota(D); di.upload(w.data(), D, s);
    hipLaunchKernelGGL(iou_cost_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, sl.p, (const int *)nullptr, T, dt.p, di.p, D, 0.f, 0.f, o.p, (const int *)nullptr, (const int *)nullptr,
                       (const int *)nullptr, (const int *)nullptr);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
static int nn_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out,
                       int euclid) {
    YDS_API_BEGIN
    using namespace yds;
    // nn_matching.py:4-53 is dimension agnostic; the kernel reduces 512-wide rows, so narrower embeddings are zero padded (zeros
    // change neither a dot product nor a norm) and wider ones are refused
    if (dim < 1 || dim > EMB) fail("nn distance: feature dimension %d outside [1, %d]", dim, EMB);
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    int budget = 1;
    for (int t = 0; t < T; ++t) {
        if (seg_offsets_host[t + 1] < seg_offsets_host[t]) fail("nn distance: segment offsets must not decrease");
        budget = std::max(budget, seg_offsets_host[t + 1] - seg_offsets_host[t]);
    }
    std::vector<float> g((size_t)T * budget * EMB, 0.f), fpad;
    std::vector<int> rows(T);
    for (int t = 0; t < T; ++t) {
        rows[t] = seg_offsets_host[t + 1] - seg_offsets_host[t];
        for (int r = 0; r < rows[t]; ++r)
            memcpy(&g[((size_t)t * budget + r) * EMB], gallery_host + ((size_t)seg_offsets_host[t] + r) * dim, (size_t)dim * 4);
    }
    if (dim != EMB) {
        fpad.assign((size_t)D * EMB, 0.f);
        for (int d = 0; d < D; ++d) memcpy(&fpad[(size_t)d * EMB], feats_host + (size_t)d * dim, (size_t)dim * 4);
        feats_host = fpad.data();
    }
    DevBuf<float> gd, fd, gn(g.size()), fn((size_t)D * EMB), o((size_t)T * D); DevBuf<int> sl, nr;
    gd.upload(g.data(), g.size(), s); fd.upload(feats_host, (size_t)D * EMB, s);
    auto v = iota(T); sl.upload(v.data(), T, s); nr.upload(rows.data(), T, s);
    const int G = T * budget;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((G + 3) / 4), dim3(256), 0, s, gd.p, (const int *)nullptr, gn.p, G, euclid ? 0 : 1);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((D + 3) / 4), dim3(256), 0, s, fd.p, (const int *)nullptr, fn.p, D, euclid ? 0 : 1);
    hipLaunchKernelGGL(appearance_cost_kernel, dim3(T, (D + 15) / 16), dim3(256), 0, s, gn.p, sl.p, nr.p, budget, fn.p, D,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, 0.f, 0, euclid, o.p, (const int *)nullptr,
                       (const int *)nullptr, (const int *)nullptr, (const int *)nullptr);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_cosine_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out) {
    return nn_min_cost(gallery_host, seg_offsets_host, T, feats_host, D, dim, out, 0);
}
int yds_euclidean_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out) {
    return nn_min_cost(gallery_host, seg_offsets_host, T, feats_host, D, dim, out, 1);
}
int yds_tracker_nms(const float *tlwh_host, const int32_t *order_host, int D, double max_overlap, int32_t *pick_host, int *n_pick) {
    YDS_API_BEGIN
    using namespace yds;
    *n_pick = 0;
    if (D == 0) return 0;
    for (int k = 0; k < D; ++k) if (order_host[k] < 0 || order_host[k] >= D) fail("tracker_nms: order[%d] = %d outside [0,%d)", k, order_host[k], D);
    hipStream_t s = g_scratch.stream();
    DevBuf<float> b; DevBuf<int> ord, pick(D), cnt(1);
    b.upload(tlwh_host, (size_t)D * 4, s); ord.upload(order_host, D, s);
    hipLaunchKernelGGL(tracker_nms_kernel, dim3(1), dim3(256), (size_t)D * sizeof(int), s, b.p, ord.p, D, max_overlap, pick.p, cnt.p);
    YDS_HIP(hipGetLastError());
    YDS_HIP(hipMemcpyAsync(n_pick, cnt.p, sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(pick_host, pick.p, (size_t)D * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}

}  // extern "C"
