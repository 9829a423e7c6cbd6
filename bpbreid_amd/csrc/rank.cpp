// Native CMC / mAP evaluation (market1501 protocol), the counterpart of the reference's only native
// component torchreid/metrics/rank_cylib/rank_cy.pyx:154-241 (compiled but never called: rank.py:205-214
// route both branches to the pure-Python loop, 23 ms/query at G=20k).  Host C++, multi-threaded over queries;
// the per-query argsort is a stable sort on (distance, gallery index), i.e. numpy's order on tie-free rows and
// a deterministic lowest-index-first order on ties (documented in tests/test_metrics.py).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <numeric>
#include <thread>
#include <vector>

#include "bpb_common.h"

extern "C" int bpb_eval_rank(const float* distmat, const int64_t* q_pids, const int64_t* g_pids, const int64_t* q_camids,
                             const int64_t* g_camids, int Q, int G, int max_rank, int nthreads, float* cmc_out,
                             double* map_out, int* num_valid_out, int32_t* indices_out)
{
    if (Q < 1 || G < 1 || max_rank < 1) return bpb_set_error(-1, "bpb_eval_rank: bad sizes");
    if (max_rank > G) max_rank = G;
    if (nthreads < 1) nthreads = 1;
    std::vector<std::vector<double>> cmc_acc(nthreads, std::vector<double>(max_rank, 0.0));
    std::vector<double> ap(Q, -1.0);
    std::atomic<int> next(0);
    auto worker = [&](int tid) {
        std::vector<int32_t> order(G);
        for (;;) {
            const int q = next.fetch_add(1);
            if (q >= Q) break;
            const float* row = distmat + (size_t)q * G;
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [row](int32_t a, int32_t b) { return row[a] < row[b]; });
            if (indices_out) std::copy(order.begin(), order.end(), indices_out + (size_t)q * G);
            // drop gallery samples with the same pid AND camid as the query (rank.py:122-125)
            long num_rel = 0, kept = 0, hits = 0;
            double ap_sum = 0.0;
            bool any = false;
            int first_hit_rank = -1;
            for (int k = 0; k < G; ++k) {
                const int32_t g = order[k];
                if (g_pids[g] == q_pids[q] && g_camids[g] == q_camids[q]) continue;
                const bool match = g_pids[g] == q_pids[q];
                ++kept;
                if (match) {
                    ++hits;
                    ++num_rel;
                    ap_sum += (double)hits / (double)kept;     // precision at each relevant position (rank.py:142-147)
                    if (!any) { any = true; first_hit_rank = (int)kept - 1; }
                }
            }
            if (!any) continue;                                  // query identity absent from gallery (rank.py:131-133)
            for (int r = first_hit_rank; r < max_rank; ++r) cmc_acc[tid][r] += 1.0;
            ap[q] = ap_sum / (double)num_rel;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    int nvalid = 0;
    double ap_total = 0.0;
    for (int q = 0; q < Q; ++q)
        if (ap[q] >= 0.0) { ++nvalid; ap_total += ap[q]; }
    if (num_valid_out) *num_valid_out = nvalid;
    if (nvalid == 0) return bpb_set_error(-2, "bpb_eval_rank: all query identities do not appear in gallery");
    for (int r = 0; r < max_rank; ++r) {
        double s = 0.0;
        for (int t = 0; t < nthreads; ++t) s += cmc_acc[t][r];
        cmc_out[r] = (float)(s / nvalid);
    }
    *map_out = ap_total / nvalid;
    return 0;
}
