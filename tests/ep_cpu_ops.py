"""Oracle-backed implementation of the four EP compute steps (test infrastructure): the same buffer
contract as the HIP kernels behind moeinf_ep_pack / moeinf_ep_expert_ffn / moeinf_ep_combine, on CPU
tensors, so moe-infinity_amd/ep.py's host logic can run under gloo without a GPU."""
import torch

from oracle import moe_ref as R


class OracleEpOps:
    def __init__(self, experts_by_layer, rank, world, top_k, num_experts, hidden):
        self.experts, self.rank, self.world, self.K, self.E, self.H = experts_by_layer, rank, world, top_k, num_experts, hidden

    def route(self, layer, x2, gate_w):
        self.sel, self.w, _ = R.route_mixtral(x2, gate_w, self.K)

    def row_elems(self):
        return self.H + 8  # bf16: 16-byte tail = 8 elements; first int32 of the tail = expert id

    @staticmethod
    def _meta(buf, H):
        """int32 view of the first 4 bytes of every row's tail."""
        return buf[:, H:H + 2].view(torch.int32).reshape(-1)

    def pack(self, x2, send, counts, cap_rows):
        T = x2.shape[0]
        H = x2.shape[1]
        send.zero_()
        meta = self._meta(send, H)
        meta.fill_(-1)
        cnt = [0] * self.world
        self.pair_pos = torch.full((T, self.K), -1, dtype=torch.int64)
        for t in range(T):  # pair order = token-major, the order the dispatch-index kernel ranks in
            for k in range(self.K):
                e = int(self.sel[t, k])
                d = e % self.world
                row = d * cap_rows + cnt[d]
                cnt[d] += 1
                send[row, :H] = x2[t]
                meta[row] = e
                self.pair_pos[t, k] = row
        if counts is not None:
            counts.copy_(torch.tensor(cnt, dtype=counts.dtype))

    def pack_compact(self, x2, send, counts):
        """rows sorted by destination rank (stable: pair order inside a destination), compact"""
        T, H = x2.shape
        meta = self._meta(send, H)
        pairs = [(int(self.sel[t, k]) % self.world, t, k) for t in range(T) for k in range(self.K)]
        order = sorted(range(len(pairs)), key=lambda i: pairs[i][0])  # Python's sort is stable
        self.pair_pos = torch.full((T, self.K), -1, dtype=torch.int64)
        cnt = [0] * self.world
        for row, i in enumerate(order):
            d, t, k = pairs[i]
            send[row, :H] = x2[t]
            meta[row] = int(self.sel[t, k])
            self.pair_pos[t, k] = row
            cnt[d] += 1
        counts.copy_(torch.tensor(cnt, dtype=counts.dtype))

    def expert_ffn_rows(self, layer, recv, y, nrows):
        self.expert_ffn(layer, recv[:nrows], y[:nrows], None)

    def expert_ffn(self, layer, recv, y, cap_rows):
        y.zero_()
        H = y.shape[1]
        meta = self._meta(recv, H)
        for e in sorted({int(v) for v in meta.tolist() if v >= 0}):
            assert e % self.world == self.rank, "received rows for an expert this rank does not own"
            rows = (meta == e).nonzero().flatten()
            y[rows] = R.expert_ffn(recv[rows][:, :H].contiguous(), self.experts[layer][e], R.MIXTRAL_DENSE_ACT_DENSE)

    def combine(self, x2, ret, out, cap_rows):
        T = x2.shape[0]
        out.zero_()
        for t in range(T):
            order = sorted(range(self.K), key=lambda k: int(self.sel[t, k]))  # ascending expert id
            acc = torch.zeros_like(out[t])
            for k in order:
                acc = acc + ret[self.pair_pos[t, k]] * self.w[t, k]  # bf16 mul then bf16 add, as mixtral.py:96-101
            out[t] = acc
