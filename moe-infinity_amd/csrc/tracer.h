// tracer.h — activation-aware expert tracer / predictor / prefetch ordering (host only).
//
// Restates moe_infinity/memory/expert_tracer.py (ExpertTracer), expert_predictor.py
// (ExpertPredictor.predict) and expert_prefetcher.py (ExpertPrefetcher.prefetch_experts ordering).
// EAM = expert activation matrix [L,E] of one sequence; the collection holds `capacity`
// historical EAMs.
//
// The reference recomputes, per sequence per layer, a [capacity,L,E] clone + normalisation +
// cosine + mean + argmin on cuda:0 (expert_tracer.py:94-125).  Here the collection side is
// normalised once, and per sequence the cosines C[i][l] = cos(EAM_l, hist_i_l) are cached and
// only column `layer` is refreshed when update_entry touches row `layer`; a predict() is then
// O(capacity*(E + L)) on the host instead of O(capacity*L*E) on the GPU plus a D2H sync.
// The result (index of the nearest EAM) is the same function:
//   cos_dist_i = 1 - mean_l cos(normalise(EAM)_l, normalise(hist_i with rows <= layer := 1e-9)_l)
// Rows <= layer are identical for every i and cannot change the argmin, so they are left out.
// Quirk kept: an all-zero historical row makes the reference's normalisation 0/0 = NaN, and
// torch.argmin returns the FIRST NaN — i.e. while the collection has empty slots the "nearest"
// EAM is the first empty slot for every layer but the last.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <unordered_map>
#include <vector>

namespace moeinf {

class Tracer {
 public:
  Tracer(int L, int E, int cap) : L_(L), E_(E), cap_(cap) {
    coll_.assign((size_t)cap * L * E, 0.f);
    chat_.assign((size_t)cap * L * E, 0.0);
    cnan_.assign((size_t)cap * L, 1);
    access_.assign(cap, 0.0);
  }
  int layers() const { return L_; }
  int experts() const { return E_; }
  int capacity() const { return cap_; }
  bool has(int64_t id) const { return entries_.count(id) != 0; }

  // expert_tracer.py:40-52 (load_trace): first n slots are the persistent trace
  void load(const float* eams, int n) {
    std::copy(eams, eams + (size_t)n * L_ * E_, coll_.begin());
    std::fill(coll_.begin() + (size_t)n * L_ * E_, coll_.end(), 0.f);
    persistent_ = n;
    for (int i = 0; i < cap_; ++i) renorm_entry(i);
    for (auto& kv : entries_) std::fill(kv.second.dirty.begin(), kv.second.dirty.end(), 1);
  }

  // expert_tracer.py:54-59
  int64_t create_entry() {
    const int64_t id = next_id_++;
    Entry& en = entries_[id];
    en.m.assign((size_t)L_ * E_, 0.0);
    en.C.assign((size_t)cap_ * L_, 0.0);
    en.dirty.assign(L_, 1);
    en.new_tokens = 0;
    return id;
  }

  // expert_tracer.py:61-76
  void finish_entry(int64_t id) {
    Entry& en = entries_[id];
    int idx = -1;
    for (int i = 0; i < cap_ && idx < 0; ++i) {
      double s = 0;
      for (int k = 0; k < L_ * E_; ++k) s += coll_[(size_t)i * L_ * E_ + k];
      if (s == 0) idx = i;
    }
    if (idx < 0) {
      double best = 1e300;
      for (int i = 0; i < cap_; ++i) {
        const double a = (i < persistent_) ? 1e9 : access_[i];
        if (a < best) { best = a; idx = i; }
      }
    }
    for (int k = 0; k < L_ * E_; ++k) coll_[(size_t)idx * L_ * E_ + k] = (float)en.m[k];
    access_[idx] = 1;
    renorm_entry(idx);
    for (auto& kv : entries_) std::fill(kv.second.dirty.begin(), kv.second.dirty.end(), 1);
    entries_.erase(id);
  }

  // expert_predictor.py:17-35: update_entry (expert_tracer.py:78-84) + find_most_similar (:94-125) + decay
  int predict(int64_t id, int layer, const int32_t* experts, int n, float* out) {
    Entry& en = entries_[id];
    for (int i = 0; i < n; ++i) en.m[(size_t)layer * E_ + experts[i]] += 1.0;
    if (layer == L_ - 1) en.new_tokens += 1;
    en.dirty[layer] = 1;
    const int nearest = find_most_similar(en, layer);
    access_[nearest] += 1;
    // expert_matrix[:layer] = 0 ; expert_matrix[l] = (expert_matrix[l] + 1e-8) * decay(l), float32 arithmetic
    const float* src = &coll_[(size_t)nearest * L_ * E_];
    for (int l = 0; l < L_; ++l) {
      const float d = (float)(-1.0 / (L_ + 1) * (double)(l - layer) + 1.0);
      for (int e = 0; e < E_; ++e) out[(size_t)l * E_ + e] = (l < layer) ? 0.f : (src[(size_t)l * E_ + e] + 1e-8f) * d;
    }
    return nearest;
  }

  // expert_prefetcher.py:42-59: (layer, expert) with score > 0 for layers >= layer, stable descending by score
  int prefetch_order(int layer, const float* matrix, int32_t* layers_out, int32_t* experts_out, float* scores_out) const {
    std::vector<int> ids;
    for (int l = layer; l < L_; ++l)
      for (int e = 0; e < E_; ++e)
        if (matrix[(size_t)l * E_ + e] > 0) ids.push_back(l * E_ + e);
    std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return matrix[a] > matrix[b]; });
    for (size_t i = 0; i < ids.size(); ++i) {
      layers_out[i] = ids[i] / E_;
      experts_out[i] = ids[i] % E_;
      if (scores_out) scores_out[i] = matrix[ids[i]];
    }
    return (int)ids.size();
  }

  void get_eam(int64_t id, double* out) { const Entry& en = entries_[id]; std::copy(en.m.begin(), en.m.end(), out); }

 private:
  struct Entry {
    std::vector<double> m;       // [L][E] EAM (float64 numpy in the reference)
    std::vector<double> C;       // [cap][L] cached cosines against the collection
    std::vector<uint8_t> dirty;  // [L] row changed since C[:, l] was computed
    int64_t new_tokens;
  };

  // normalise one historical EAM the way find_most_similar does: row / row.sum() in float32
  // (0/0 -> NaN), then unit length with eps 1e-6 in float64 (nn.CosineSimilarity(dim=2, eps=1e-6))
  void renorm_entry(int i) {
    for (int l = 0; l < L_; ++l) {
      const float* r = &coll_[((size_t)i * L_ + l) * E_];
      float s = 0.f;
      for (int e = 0; e < E_; ++e) s += r[e];
      double* c = &chat_[((size_t)i * L_ + l) * E_];
      if (s == 0.f) {
        cnan_[(size_t)i * L_ + l] = 1;
        continue;
      }
      cnan_[(size_t)i * L_ + l] = 0;
      double nrm = 0;
      for (int e = 0; e < E_; ++e) { const double v = (double)(r[e] / s); c[e] = v; nrm += v * v; }
      nrm = std::max(sqrt(nrm), 1e-6);
      for (int e = 0; e < E_; ++e) c[e] /= nrm;
    }
  }

  int find_most_similar(Entry& en, int layer) {
    std::vector<double> mh(E_);
    for (int l = layer + 1; l < L_; ++l) {
      if (!en.dirty[l]) continue;
      const double* m = &en.m[(size_t)l * E_];
      double s = 0;
      for (int e = 0; e < E_; ++e) s += m[e];
      double nrm = 0;
      for (int e = 0; e < E_; ++e) { mh[e] = (s == 0) ? 0.0 : m[e] / s; nrm += mh[e] * mh[e]; }  // nan_to_num
      nrm = std::max(sqrt(nrm), 1e-6);
      for (int e = 0; e < E_; ++e) mh[e] /= nrm;
      for (int i = 0; i < cap_; ++i) {
        const double* c = &chat_[((size_t)i * L_ + l) * E_];
        double d = 0;
        for (int e = 0; e < E_; ++e) d += mh[e] * c[e];
        en.C[(size_t)i * L_ + l] = d;
      }
      en.dirty[l] = 0;
    }
    // first NaN wins (torch.argmin); otherwise smallest 1 - mean(cos)
    for (int i = 0; i < cap_; ++i)
      for (int l = layer + 1; l < L_; ++l)
        if (cnan_[(size_t)i * L_ + l]) return i;
    int best = 0;
    double best_d = 1e300;
    for (int i = 0; i < cap_; ++i) {
      double s = 0;
      for (int l = layer + 1; l < L_; ++l) s += en.C[(size_t)i * L_ + l];
      const double dist = 1.0 - s / L_;
      if (dist < best_d) { best_d = dist; best = i; }
    }
    return best;
  }

  int L_, E_, cap_, persistent_ = 0;
  int64_t next_id_ = 1;
  std::vector<float> coll_;
  std::vector<double> chat_;
  std::vector<uint8_t> cnan_;
  std::vector<double> access_;
  std::unordered_map<int64_t, Entry> entries_;
};

}  // namespace moeinf
