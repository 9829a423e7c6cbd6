// k-reciprocal re-ranking (Zhong et al., CVPR 2017) of a query x gallery distance matrix: the native counterpart of
// torchreid/utils/rerank.py:30-117 (a pure numpy/Python loop over all Q+G samples, called from engine.py:433-437).
// Host C++, multi-threaded over samples.  Same arithmetic in fp32 and the same set semantics as the reference:
//   - squared distances, normalised by the column maximum, transposed (rerank.py:42-46);
//   - initial ranking = ascending order of every row; only the first k1+1 entries are ever used, so a partial sort;
//     ties are broken by the lower index (numpy's unstable argsort leaves them unspecified);
//   - k-reciprocal sets R(i, k1), expanded by R(c, round(k1/2)) of every member c whose set overlaps R(i, k1) in more than
//     2/3 of its elements (:55-78), Gaussian-kernel weights normalised to one (:81-82);
//   - local query expansion: V[i] <- mean of the rows of i's k2 nearest neighbours (:84-89);
//   - Jaccard distance through the inverted index (:91-107), blended with the original distance (:109).
// V is kept as sorted sparse rows (a few dozen non-zeros each) instead of the reference's dense (Q+G)^2 array.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <thread>
#include <vector>

#include "bpb_common.h"

namespace {
struct SparseRow {
    std::vector<int32_t> idx;
    std::vector<float> val;
};

template <class F>
void parallel_for(int n, int nthreads, F&& fn)
{
    std::atomic<int> next(0);
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            fn(i);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
}
}   // namespace

extern "C" int bpb_re_ranking(const float* q_g, const float* q_q, const float* g_g, int Q, int G, int k1, int k2,
                              float lambda_value, int nthreads, float* final_dist)
{
    if (Q < 1 || G < 1) return bpb_set_error(-1, "bpb_re_ranking: bad sizes");
    const int N = Q + G;
    if (k1 < 1 || k2 < 1 || k1 + 1 > N || k2 > N) return bpb_set_error(-1, "bpb_re_ranking: k1=%d k2=%d need k1 + 1 <= Q + G = %d", k1, k2, N);
    if (nthreads < 1) nthreads = 1;
    auto raw = [&](int i, int j) -> float {      // [[q_q, q_g], [q_g^T, g_g]]
        if (i < Q) return j < Q ? q_q[(size_t)i * Q + j] : q_g[(size_t)i * G + (j - Q)];
        return j < Q ? q_g[(size_t)j * G + (i - Q)] : g_g[(size_t)(i - Q) * G + (j - Q)];
    };
    // od[i][j] = raw[j][i]^2 / max_k raw[k][i]^2
    std::vector<float> od((size_t)N * N);
    parallel_for(N, nthreads, [&](int i) {
        float mx = 0.f;
        bool first = true;
        float* row = od.data() + (size_t)i * N;
        for (int j = 0; j < N; ++j) {
            const float r = raw(j, i);
            const float d2 = r * r;
            row[j] = d2;
            if (first || d2 > mx) { mx = d2; first = false; }
        }
        for (int j = 0; j < N; ++j) row[j] = 1.f * row[j] / mx;
    });
    // initial ranking, first kk = max(k1 + 1, k2) entries of every row
    const int kh = (int)std::nearbyint(k1 / 2.0) + 1;            // np.around: half to even
    const int kk = std::max(k1 + 1, k2);
    std::vector<int32_t> rank((size_t)N * kk);
    parallel_for(N, nthreads, [&](int i) {
        const float* row = od.data() + (size_t)i * N;
        std::vector<int32_t> order(N);
        std::iota(order.begin(), order.end(), 0);
        auto less = [row](int32_t a, int32_t b) { return row[a] < row[b] || (row[a] == row[b] && a < b); };
        std::partial_sort(order.begin(), order.begin() + kk, order.end(), less);
        std::copy(order.begin(), order.begin() + kk, rank.begin() + (size_t)i * kk);
    });
    auto reciprocal = [&](int i, int k, std::vector<int32_t>& out) {   // forward neighbours f (in rank order) with i in top-k of f
        out.clear();
        const int32_t* fw = rank.data() + (size_t)i * kk;
        for (int p = 0; p < k; ++p) {
            const int32_t* bw = rank.data() + (size_t)fw[p] * kk;
            for (int q = 0; q < k; ++q)
                if (bw[q] == i) { out.push_back(fw[p]); break; }
        }
    };
    std::vector<SparseRow> V(N);
    parallel_for(N, nthreads, [&](int i) {
        std::vector<int32_t> rec, cand, expansion, a, b;
        reciprocal(i, k1 + 1, rec);
        expansion = rec;
        a = rec;
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
        for (int32_t c : rec) {
            reciprocal(c, kh, cand);
            b = cand;
            std::sort(b.begin(), b.end());
            b.erase(std::unique(b.begin(), b.end()), b.end());
            size_t common = 0;
            for (size_t x = 0, y = 0; x < a.size() && y < b.size();) {
                if (a[x] == b[y]) { ++common; ++x; ++y; }
                else if (a[x] < b[y]) ++x;
                else ++y;
            }
            if ((double)common > 2. / 3 * (double)cand.size()) expansion.insert(expansion.end(), cand.begin(), cand.end());
        }
        std::sort(expansion.begin(), expansion.end());
        expansion.erase(std::unique(expansion.begin(), expansion.end()), expansion.end());
        SparseRow& r = V[i];
        r.idx = expansion;
        r.val.resize(expansion.size());
        const float* row = od.data() + (size_t)i * N;
        float sum = 0.f;
        for (size_t t = 0; t < expansion.size(); ++t) {
            r.val[t] = std::exp(-row[expansion[t]]);
            sum += r.val[t];
        }
        for (float& v : r.val) v = 1.f * v / sum;
    });
    if (k2 != 1) {
        std::vector<SparseRow> Vq(N);
        parallel_for(N, nthreads, [&](int i) {
            std::vector<float> acc;           // dense scratch over the union of the k2 supports would be N wide; merge instead
            std::vector<int32_t> uni;
            const int32_t* nb = rank.data() + (size_t)i * kk;
            for (int t = 0; t < k2; ++t) uni.insert(uni.end(), V[nb[t]].idx.begin(), V[nb[t]].idx.end());
            std::sort(uni.begin(), uni.end());
            uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
            acc.assign(uni.size(), 0.f);
            for (int t = 0; t < k2; ++t) {    // rows added in neighbour order, like numpy's reduction over axis 0
                const SparseRow& s = V[nb[t]];
                size_t u = 0;
                for (size_t e = 0; e < s.idx.size(); ++e) {
                    while (uni[u] != s.idx[e]) ++u;
                    acc[u] += s.val[e];
                }
            }
            SparseRow& r = Vq[i];
            for (size_t u = 0; u < uni.size(); ++u) {
                const float m = acc[u] / (float)k2;
                if (m != 0.f) { r.idx.push_back(uni[u]); r.val.push_back(m); }
            }
        });
        V.swap(Vq);
    }
    // inverted index: for every column the rows (ascending) whose V is non-zero there, with the value
    std::vector<std::vector<std::pair<int32_t, float>>> inv(N);
    for (int r = 0; r < N; ++r)
        for (size_t e = 0; e < V[r].idx.size(); ++e) inv[V[r].idx[e]].emplace_back(r, V[r].val[e]);
    parallel_for(Q, nthreads, [&](int i) {
        std::vector<float> tmin(N, 0.f);
        const SparseRow& vi = V[i];
        for (size_t e = 0; e < vi.idx.size(); ++e)
            for (const auto& rv : inv[vi.idx[e]]) tmin[rv.first] = tmin[rv.first] + std::min(vi.val[e], rv.second);
        const float* row = od.data() + (size_t)i * N;
        for (int g = 0; g < G; ++g) {
            const float t = tmin[Q + g];
            const float jac = 1.f - t / (2.f - t);
            final_dist[(size_t)i * G + g] = jac * (1.f - lambda_value) + row[Q + g] * lambda_value;
        }
    });
    return 0;
}
