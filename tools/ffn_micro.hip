// ffn_micro.hip — the REAL FFN kernels (csrc/kernels.hip, included) under a back-to-back launch harness with rotating
// weight sets, DeepSeek-V2-Lite decode shapes (B=1: 6 routed experts + the shared expert).  Isolates kernel time from
// the engine: what does each stage cost, what do the fused combine / separate slot allocations add?
// Build: hipcc --offload-arch=gfx950 -O3 -I moe-infinity_amd/csrc tools/ffn_micro.hip
#include "../moe-infinity_amd/csrc/kernels.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace moeinf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int H = 2048, F = 1408, Fs = 2816, E = 64, K = 6;
  const bool with_shared = argc > 1 ? atoi(argv[1]) != 0 : true;
  const int nact = K + (with_shared ? 1 : 0);
  auto al = [](int64_t v) { return (v + 4095) / 4096 * 4096; };
  const int64_t t_fh = tiled_bytes(F, H, DT_BF16), t_hf = tiled_bytes(H, F, DT_BF16), t_fsh = tiled_bytes(Fs, H, DT_BF16), t_hfs = tiled_bytes(H, Fs, DT_BF16);
  const int64_t blob = al(t_fh) * 2 + al(t_hf), blob_sh = al(t_fsh) * 2 + al(t_hfs);
  const int64_t set_bytes = blob * K + blob_sh;
  const int nset = (int)((700ll << 20) / set_bytes) + 1;
  printf("blob %.1f MB, shared %.1f MB, %d rotating sets, shared expert %s\n", blob / 1e6, blob_sh / 1e6, nset, with_shared ? "on" : "off");
  std::vector<char*> sets(nset);
  for (auto& s : sets) { CK(hipMalloc(&s, set_bytes)); CK(hipMemset(s, 0, set_bytes)); }
  uint64_t* wptr; CK(hipMalloc(&wptr, (size_t)nset * (E + 1) * 8));
  std::vector<uint64_t> hw((size_t)nset * (E + 1), 0);
  for (int s = 0; s < nset; ++s) { for (int k = 0; k < K; ++k) hw[(size_t)s * (E + 1) + 3 + 7 * k] = (uint64_t)(sets[s] + blob * k); hw[(size_t)s * (E + 1) + E] = (uint64_t)(sets[s] + blob * K); }
  CK(hipMemcpy(wptr, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
  int32_t *active, *n_active, *counts, *offsets, *slot_token, *miss, *arrive, *topk_idx, *pair_slot, *pair_order;
  float* topk_w;
  CK(hipMalloc(&active, (E + 1) * 4)); CK(hipMalloc(&n_active, 4)); CK(hipMalloc(&counts, (E + 1) * 4)); CK(hipMalloc(&offsets, (E + 2) * 4));
  CK(hipMalloc(&slot_token, 64 * 4)); CK(hipMalloc(&miss, 4)); CK(hipMalloc(&arrive, 4096)); CK(hipMemset(arrive, 0, 4096)); CK(hipMemset(miss, 0, 4));
  CK(hipMalloc(&topk_idx, 64)); CK(hipMalloc(&pair_slot, 64)); CK(hipMalloc(&pair_order, 64)); CK(hipMalloc(&topk_w, 64));
  std::vector<int32_t> ha(E + 1, 0), hc(E + 1, 0), ho(E + 2, 0), hst(64, 0), hti(K), hps(K), hpo(K);
  std::vector<float> htw(K, 0.16f);
  int row = 0;
  for (int k = 0; k < K; ++k) { ha[k] = 3 + 7 * k; hc[3 + 7 * k] = 1; hti[k] = 3 + 7 * k; hps[k] = k; hpo[k] = k; }
  if (with_shared) { ha[K] = E; hc[E] = 1; }
  for (int e = 0; e <= E; ++e) { ho[e] = row; row += hc[e]; }
  ho[E + 1] = row;
  CK(hipMemcpy(active, ha.data(), (E + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(counts, hc.data(), (E + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(offsets, ho.data(), (E + 2) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(slot_token, hst.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(n_active, &nact, 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(topk_idx, hti.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(pair_slot, hps.data(), K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pair_order, hpo.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(topk_w, htw.data(), K * 4, hipMemcpyHostToDevice));
  void *x, *h, *y, *out;
  CK(hipMalloc(&x, H * 2)); CK(hipMemset(x, 0, H * 2)); CK(hipMalloc(&h, 8 * Fs * 2)); CK(hipMemset(h, 0, 8 * Fs * 2)); CK(hipMalloc(&y, 8 * H * 2)); CK(hipMalloc(&out, H * 2));
  FfnStage s1, s2;
  memset((void*)&s1, 0, sizeof s1);
  s1.active = active; s1.n_active = n_active; s1.counts = counts; s1.offsets = offsets; s1.miss_flag = miss; s1.n_active_host = -1; s1.E = E; s1.dtype = DT_BF16;
  s2 = s1;
  s1.K = H; s1.R = F; s1.K_sh = H; s1.R_sh = Fs; s1.in = x; s1.ld_in = H; s1.row_map = slot_token; s1.out = h; s1.ld_out = Fs;
  s1.off_a = 0; s1.off_b = al(t_fh); s1.off_a_sh = 0; s1.off_b_sh = al(t_fsh); s1.epi = EPI_GATED_SILU;
  s2.K = F; s2.R = H; s2.K_sh = Fs; s2.R_sh = H; s2.in = h; s2.ld_in = Fs; s2.out = y; s2.ld_out = H; s2.off_a = 2 * al(t_fh); s2.off_a_sh = 2 * al(t_fsh); s2.epi = EPI_NONE;
  FfnStage s2f = s2;
  s2f.fuse_combine = 1; s2f.tile_done = arrive;
  CombineArgs& ca = s2f.comb;
  memset((void*)&ca, 0, sizeof ca);
  ca.x = x; ca.y = y; ca.out = out; ca.topk_idx = topk_idx; ca.topk_w = topk_w; ca.pair_slot = pair_slot; ca.pair_order = pair_order;
  ca.y_shared = with_shared ? y : nullptr; ca.shared_offsets = with_shared ? offsets : nullptr; ca.shared_E = E; ca.T = 1; ca.H = H; ca.K = K; ca.kind = 1; ca.dtype = DT_BF16;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 400;
  auto run = [&](const char* name, double mb, auto launch) {
    for (int i = 0; i < 20; ++i) launch(i % nset);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch(i % nset);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    printf("%-58s %7.2f us  %7.1f GB/s\n", name, us, mb * 1e6 / us / 1e3);
  };
  const double mb1 = (2.0 * F * H * 2 * K + (with_shared ? 2.0 * Fs * H * 2 : 0)) / 1e6, mb2 = (1.0 * F * H * 2 * K + (with_shared ? 1.0 * Fs * H * 2 : 0)) / 1e6;
  auto L = [&](FfnStage st, int s) { st.wptr = wptr + (size_t)s * (E + 1); CK(launch_ffn_stage(st, nact, 1, nullptr)); };
  run("stage 1 (gate+up, fused silu*mul)", mb1, [&](int s) { L(s1, s); });
  run("stage 2 (down)", mb2, [&](int s) { L(s2, s); });
  run("stage 2 + fused combine", mb2, [&](int s) { L(s2f, s); });
  run("stage 1 then stage 2+combine (pair)", mb1 + mb2, [&](int s) { L(s1, s); L(s2f, s); });
  return 0;
}
