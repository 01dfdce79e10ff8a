"""Scalers wider than the fused kernels hold (hidden or metadata width > 64, or more hidden layers with per-image layers than one
fused launch takes): the layer-by-layer path of the engine on the GEMM kernels of csrc/wide_gemm.hip, as a mixin of `ElboEngine`
(split out of careless_amd/engine.py in round 4).  Reference: `MetadataScaler.call` (careless/models/scaling/nn.py:55-68, 92-120),
`ImageLayer` (scaling/image.py:66-125) and the tape gradient of both (models/merging/variational.py:197-202).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from careless_amd._lib import check, ptr
from careless_amd.obs import ObsData


def image_tiles(rel_seg: np.ndarray, device) -> torch.Tensor:
    """(group, first row) pairs covering every group's rows in 128-row pieces: the x-blocks of cl_wide_image_forward_tiles /
    _dgrad_tiles (per-image layers wider than 128).  `rel_seg` = the groups' row starts relative to the call's first row."""
    rel = np.asarray(rel_seg, dtype=np.int64)
    n = np.maximum(0, -(-(rel[1:] - rel[:-1]) // 128))                      # pieces per group
    g = np.repeat(np.arange(len(n)), n)
    first = np.repeat(np.concatenate([[0], np.cumsum(n)[:-1]]), n)
    row = rel[:-1][g] + 128 * (np.arange(int(n.sum())) - first)
    return torch.as_tensor(np.stack([g, row], axis=1).astype(np.int32).reshape(-1), device=device)


class WidePath:
    # -- scalers wider than 64 ---------------------------------------------------------------------------------------------
    WIDE_BUDGET = 4 << 30     # bytes of activation storage a row chunk may take ((L + 2) buffers of chunk x ld floats), and never more
                              # than a quarter of the free device memory: longer chunks = fewer launches and a better-balanced last round of row blocks

    def _wide_setup(self):
        """Work buffers of the unfused path (csrc/wide_gemm.hip): one activation buffer per hidden layer (Dense + per-image) + two
        gradient buffers of `chunk` rows, the per-layer weight-gradient partials.  The chunk is sized from the layer count and
        width so the buffers stay inside WIDE_BUDGET (and, with it, inside the device memory next to the shard)."""
        if getattr(self, "_wide", None) is not None:
            return self._wide
        lib, dev = self.lib, self.device
        ldw = int(lib.cl_wide_ld(self.w))
        nh = self.L + (self.imgl.n_image_layers if self.imgl is not None else 0)
        per_row = 4 * (nh + 2) * ldw
        free = torch.cuda.mem_get_info(dev)[0]
        budget = min(self.WIDE_BUDGET, max(free // 4, 64 << 20))
        chunk = max(128, min(budget // per_row, 1 << 21) // 128 * 128)
        self._wide = dict(ldw=ldw, chunk=chunk, nh=nh, rows=0, acts=[], dz=[])
        self._wide_buffers(chunk)
        return self._wide

    def _wide_buffers(self, rows: int):
        """(Re)allocate the row buffers for chunks of up to `rows` rows (an image larger than the default chunk needs its own size)."""
        W, lib, dev = self._wide, self.lib, self.device
        if rows <= W["rows"]:
            return
        ldw = W["ldw"]
        W["acts"] = [torch.zeros(rows * ldw, dtype=torch.float32, device=dev) for _ in range(max(2, W["nh"]))]
        W["dz"] = [torch.zeros(rows * ldw, dtype=torch.float32, device=dev) for _ in range(2)]
        W["nsplit"], W["nblk"] = int(lib.cl_wide_wgrad_splits(rows)), int(lib.cl_wide_head_blocks(rows))
        pmax = max(self.w * self.d + self.w, self.w * self.w + self.w)
        # (the fused dgrad + first-layer weight gradient writes one partial per workgroup: more of them, each smaller)
        W["wpart"] = torch.empty(max(W["nsplit"] * pmax, int(lib.cl_wide_dgrad_wgrad0_parts(rows)) * (self.w * self.d + self.w)),
                                 dtype=torch.float32, device=dev)
        W["hpart"] = torch.empty(W["nblk"] * (2 * self.w + 2), dtype=torch.float32, device=dev)
        W["rows"] = rows

    def _wide_chunks(self, obs: ObsData):
        """Row chunks [a, b) of `obs` for the layer-by-layer path.  With per-image layers a chunk holds whole images (their rows are
        consecutive: ObsData sort_images) and carries (first image, device array of the images' row starts relative to a)."""
        if getattr(obs, "wide_chunks", None) is not None:
            return obs.wide_chunks
        W = self._wide_setup()
        chunks = []
        if self.imgl is None:
            chunks = [(a, min(obs.N, a + W["chunk"]), 0, None) for a in range(0, obs.N, W["chunk"])]
        else:
            seg, M = obs.img_seg, len(obs.img_seg) - 1
            m0 = 0
            while m0 < M:
                m1 = m0 + 1
                while m1 < M and seg[m1 + 1] - seg[m0] <= W["chunk"]:
                    m1 += 1
                if seg[m1] > seg[m0]:
                    rel = torch.as_tensor((seg[m0:m1 + 1] - seg[m0]).astype(np.int32), device=self.device)
                    chunks.append((int(seg[m0]), int(seg[m1]), m0, rel))
                    if self.w > 128:            # wider than the grouped streaming kernel holds: the tiled kernel's list of row pieces
                        if getattr(obs, "wide_tiles", None) is None:
                            obs.wide_tiles = {}
                        obs.wide_tiles[int(seg[m0])] = image_tiles(seg[m0:m1 + 1] - seg[m0], self.device)
                m0 = m1
            self._wide_buffers(max(b - a for a, b, _, _ in chunks))
        obs.wide_chunks = chunks
        return chunks

    def _wide_layers(self):
        """(offset of Wt, offset of b, fan-in) of every Dense layer inside the scaler's flat W^T layout, then the head's offset."""
        out, off, fan_in = [], 0, self.d
        for _ in range(self.L):
            out.append((off, off + self.w * fan_in, fan_in))
            off += self.w * fan_in + self.w
            fan_in = self.w
        return out, off

    def _imgl_ptrs(self, flat: torch.Tensor, k: int, m0: int):
        """Device pointers of (kernel, bias) of image m0 in per-image layer k inside `flat` (params or grads): the layer's block is
        [W: n_images x (w x w) | b: n_images x w] (include/careless_hip.h)."""
        M, w = self.imgl.max_images, self.w
        base = flat.data_ptr() + 4 * (self.layout.off_imgl + k * M * (w * w + w))
        return base + 4 * m0 * w * w, base + 4 * (M * w * w + m0 * w)

    WIDE_KEEP_BUDGET = 64 << 30     # bytes of activations of a whole observation set the forward pass may keep for the backward pass
    # The fusions of round 4, each with the separate launches it replaced still behind it (the fallback of shapes outside the fused
    # kernels' envelopes).  Class attributes, not environment switches: the A/Bs are closed (NOTEBOOK.md), one parity test flips them
    # to hold the fused step against the separate launches (tests/test_gpu_parity.py).
    FUSE_PRE = True          # first Dense layer recomputed instead of stored (cl_wide_dense2_forward / _dgrad_pre / _wgrad_pre)
    FUSE_WG0 = True          # first layer's weight gradient inside the second layer's dgrad (cl_wide_dense_dgrad_pre_wgrad0)
    FUSE_HEADB = True        # the Dense(2) head's backward pass inside the top layer's weight gradient and dgrad
    FUSE_LIK = True          # the slot likelihood in the top layer's forward epilogue (cl_wide_dense_forward_head_lik)

    def _wide_pre(self) -> bool:
        """The first Dense layer is recomputed instead of stored (cl_wide_dense2_forward / _dgrad_pre / _wgrad_pre): at least two Dense
        layers, at most 15 metadata columns, hidden width up to 128."""
        return self.FUSE_PRE and self.L >= 2 and bool(self.lib.cl_wide_pre_supported(self.d, self.w))

    def _wide_head_bwd(self) -> bool:
        """The Dense(2) head's backward pass runs inside the top Dense layer's weight gradient and dgrad (cl_wide_dense_wgrad_head /
        _dgrad_head) instead of a launch of its own that writes dZ_L: Dense-only scalers whose top layer is a square one of the streaming
        kernel's widths and not the layer fed by the recomputed first one."""
        return (self.FUSE_HEADB and self.imgl is None and self.L >= 2 and not (self._wide_pre() and self.L == 2) and
                bool(self.lib.cl_wide_head_bwd_supported(self.w, self.w)))

    def _wide_keep_all(self, obs: ObsData):
        """Per-layer activation buffers over ALL rows of `obs` (list of tensors), or None when they do not fit: then the backward pass
        recomputes each chunk's forward into the chunk-sized buffers."""
        W = self._wide_setup()
        need = 4 * (W["nh"] - (1 if self._wide_pre() else 0)) * W["ldw"] * obs.N
        have = getattr(obs, "wide_full", None)
        if have is not None:
            return have
        free = torch.cuda.mem_get_info(self.device)[0]
        if need > min(self.WIDE_KEEP_BUDGET, free // 4):
            return None
        # (with the first layer recomputed its output has no buffer)
        obs.wide_full = [None if (l == 0 and self._wide_pre()) else torch.zeros(obs.N * W["ldw"], dtype=torch.float32, device=self.device)
                         for l in range(W["nh"])]
        return obs.wide_full

    def _wide_forward(self, obs: ObsData, chunk, keep: bool, st, full=None, head=None, lik=None):
        """Hidden layers on one row chunk of `obs`: layer l's output lands in acts[l] when `keep` (else two buffers alternate), or
        in the chunk's rows of the whole-set buffers `full`; returns the (buffer, ld) pairs of h_0 .. h_(L + K)."""
        a, b, m0, seg = chunk
        lib, W = self.lib, self._wide_setup()
        n, ldw, base = b - a, W["ldw"], self.params.data_ptr() + 4 * self.layout.off_mlp
        layers, _ = self._wide_layers()
        sf, leak = ptr(self.stop_flag), self.mlp.leakiness
        dst_of = (lambda l: full[l].data_ptr() + 4 * a * ldw) if full is not None else (lambda l: W["acts"][l if keep else l & 1].data_ptr())
        hs = [(obs.meta_rm.data_ptr() + 4 * a * obs.meta_ld, obs.meta_ld)]
        self._head_fused = self._lik_fused = False
        pre = self._wide_pre()
        for l, (ow, ob, fan_in) in enumerate(layers):
            if pre and l == 0:
                hs.append((None, ldw))          # h_0 is never stored: layer 1's launch makes it from the metadata
                continue
            dst = dst_of(l)
            if pre and l == 1:
                (ow0, ob0, d0) = layers[0]
                with_head = head is not None and self.L == 2 and self.imgl is None
                off_head, loc_ptr, sig_ptr = head[:3] if with_head else (0, None, None)
                check(lib.cl_wide_dense2_forward(hs[0][0], hs[0][1], d0, base + 4 * ow0, base + 4 * ob0, base + 4 * ow, base + 4 * ob, n, self.w, self.w,
                                                 leak, dst, ldw, (base + 4 * off_head) if with_head else None, self.bij_kind, self.mlp.epsilon,
                                                 loc_ptr, sig_ptr, sf, st), "cl_wide_dense2_forward")
                self._head_fused = with_head
            elif head is not None and l == self.L - 1 and self.imgl is None and fan_in <= 128 and self.w <= 128:
                # the top layer carries the Dense(2) head in its epilogue: (loc, sigma) come out of the same pass
                off_head, loc_ptr, sig_ptr = head[:3]
                dsd_ptr = head[3] if len(head) > 3 else None       # d sigma / d raw per row, for the head backward fused into this layer's backward
                rc = -2
                if lik is not None:
                    # ... and the slot likelihood of the chunk's rows too: (loc, sigma) never wait in memory for a launch of their own
                    rc = lib.cl_wide_dense_forward_head_lik(hs[-1][0], hs[-1][1], base + 4 * ow, base + 4 * ob, n, fan_in, self.w, leak, dst, ldw,
                                                            base + 4 * off_head, self.bij_kind, self.mlp.epsilon, loc_ptr, sig_ptr, dsd_ptr,
                                                            C.byref(lik), sf, st)
                    if rc != -2:
                        check(rc, "cl_wide_dense_forward_head_lik")
                        self._lik_fused = True
                if rc == -2:
                    check(lib.cl_wide_dense_forward_head(hs[-1][0], hs[-1][1], base + 4 * ow, base + 4 * ob, n, fan_in, self.w, leak, dst, ldw,
                                                         base + 4 * off_head, self.bij_kind, self.mlp.epsilon, loc_ptr, sig_ptr, dsd_ptr, sf, st),
                          "cl_wide_dense_forward_head")
                self._head_fused = True
            else:
                check(lib.cl_wide_dense_forward(hs[-1][0], hs[-1][1], base + 4 * ow, base + 4 * ob, n, fan_in, self.w, leak, 1,
                                                dst, ldw, sf, st), "cl_wide_dense_forward")
            hs.append((dst, ldw))
        for k in range(self.imgl.n_image_layers if self.imgl is not None else 0):       # image.py:116-125
            l = self.L + k
            dst = dst_of(l)
            wk, bk = self._imgl_ptrs(self.params, k, m0)
            if self.w > 128:
                tl = obs.wide_tiles[a]
                check(lib.cl_wide_image_forward_tiles(hs[-1][0], hs[-1][1], wk, bk, ptr(seg), ptr(tl), tl.numel() // 2, self.w, leak, dst, ldw, sf, st),
                      "cl_wide_image_forward_tiles")
            else:
                check(lib.cl_wide_image_forward(hs[-1][0], hs[-1][1], wk, bk, ptr(seg), seg.numel() - 1, n, self.w, leak, dst, ldw, sf, st),
                      "cl_wide_image_forward")
            hs.append((dst, ldw))
        return hs

    def _data_term_wide(self, obs: ObsData, step: int, eta, ipred_out, st):
        """Hidden / metadata width > 64 (or more hidden layers with per-image layers than one fused launch holds): unfused scaler on
        the GEMM kernels of csrc/wide_gemm.hip.  Forward (row chunks; the top layer carries the Dense(2) head) -> (loc, sigma) per row ->
        the slot likelihood (one launch when every row is its own slot, the three Laue launches otherwise) -> dL/d(loc, sigma) -> per
        chunk: forward again unless the activations were kept; the head's backward pass inside the top layer's weight gradient and
        dgrad (or its own launch outside their envelope), then weight gradient and dgrad layer by layer, top down, the first layer's
        weight gradient inside the second layer's dgrad when the first layer is recomputed.  6 (activations kept) or 8 P_mm flops per
        observation; every product in exact fp32."""
        lib, lay, W = self.lib, self.layout, self._wide_setup()
        chunks = self._wide_chunks(obs)
        ma = self._mlp_args(step, eta, ipred_out, obs)
        layers, off_head = self._wide_layers()
        pbase = self.params.data_ptr() + 4 * lay.off_mlp
        gbase = self.grads.data_ptr() + 4 * lay.off_mlp
        sf, leak, w, ldw = ptr(self.stop_flag), self.mlp.leakiness, self.w, W["ldw"]
        K = self.imgl.n_image_layers if self.imgl is not None else 0
        # The activations of ALL rows are kept by the forward pass when they fit (WIDE_KEEP_BUDGET, a quarter of the free memory at
        # most: 15 GB for 10 M rows of a 3 x 128 scaler on a 288-GB device) and the backward pass starts from them -- 6 P_mm flops per
        # observation.  Otherwise the buffers hold one chunk at a time and each chunk's forward is recomputed when its turn in the
        # backward pass comes (8 P_mm).
        full = self._wide_keep_all(obs)
        kept = []
        headb = self._wide_head_bwd()
        if headb and getattr(obs, "wide_dsd", None) is None:
            obs.wide_dsd = torch.empty(obs.N, dtype=torch.float32, device=self.device)
        # The slot likelihood rides in the top layer's forward epilogue when a production step asks for nothing else of it (rows that are
        # their own slot, in-kernel noise, no predictions out, no Evans-2011 terms, not the deterministic mode); the library decides by
        # shape (-2), the same for every chunk
        want_lik = (self.FUSE_LIK and obs.harmonic_id is None and eta is None and ipred_out is None and not self.ev11 and not self.deterministic)
        lik_fused = []
        for ch in chunks:
            a, b = ch[0], ch[1]
            head = (off_head, obs.laue_loc.data_ptr() + 4 * a, obs.laue_sig.data_ptr() + 4 * a)
            if headb:
                head = head + (obs.wide_dsd.data_ptr() + 4 * a,)
            lik = self._slot_args(ma, obs, step, None, None, a, b - a) if want_lik else None
            hs = (self._wide_forward(obs, ch, True, st, full=full, head=head, lik=lik) if full is not None
                  else self._wide_forward(obs, ch, False, st, head=head, lik=lik))
            kept.append(hs)
            lik_fused.append(self._lik_fused)
            if not self._head_fused:
                check(lib.cl_wide_head_forward(hs[-1][0], hs[-1][1], pbase + 4 * off_head, b - a, w, self.bij_kind, self.mlp.epsilon,
                                               obs.laue_loc.data_ptr() + 4 * a, obs.laue_sig.data_ptr() + 4 * a, sf, st), "cl_wide_head_forward")
        if not all(lik_fused):
            assert not any(lik_fused)
            self._slot_likelihood(ma, obs, step, eta, ipred_out, st)
        for ic, ch in enumerate(chunks):
            a, b, m0, seg = ch
            n = b - a
            hs = kept[ic] if full is not None else self._wide_forward(obs, ch, True, st)
            dz, dzn = W["dz"]
            nsplit = min(W["nsplit"], int(lib.cl_wide_wgrad_splits(n)))
            if headb:
                # the head's backward pass rides on the top layer's two backward kernels: dZ_L is made from h_L where they read it
                l = self.L - 1
                ow, ob, fan_in = layers[l]
                hd = (hs[-1][0], hs[-1][1], pbase + 4 * off_head, obs.laue_dO.data_ptr() + 8 * a, obs.wide_dsd.data_ptr() + 4 * a)
                check(lib.cl_wide_dense_wgrad_head(*hd, leak, hs[l][0], hs[l][1], n, w, fan_in, ptr(W["wpart"]), ptr(W["hpart"]), nsplit, sf, st),
                      "cl_wide_dense_wgrad_head")
                check(lib.cl_reduce_partials(ptr(W["wpart"]), nsplit, w * fan_in + w, gbase + 4 * ow, sf, st), "cl_reduce_partials")
                check(lib.cl_reduce_partials(ptr(W["hpart"]), nsplit, 2 * w + 2, gbase + 4 * off_head, sf, st), "cl_reduce_partials")
                check(lib.cl_wide_dense_dgrad_head(*hd, pbase + 4 * ow, n, w, fan_in, hs[l][0], hs[l][1], leak, ptr(dz), ldw, sf, st),
                      "cl_wide_dense_dgrad_head")
            else:
                nblk = min(W["nblk"], int(lib.cl_wide_head_blocks(n)))
                check(lib.cl_wide_head_backward(hs[-1][0], hs[-1][1], pbase + 4 * off_head, obs.laue_dO.data_ptr() + 8 * a, n, w, self.bij_kind,
                                                self.mlp.epsilon, leak, ptr(dz), ldw, ptr(W["hpart"]), nblk, sf, st), "cl_wide_head_backward")
                check(lib.cl_reduce_partials(ptr(W["hpart"]), nblk, 2 * w + 2, gbase + 4 * off_head, sf, st), "cl_reduce_partials")
            for k in range(K - 1, -1, -1):          # per-image layers: each image's gradient is written once (its rows sit in one chunk)
                l = self.L + k
                gw, gb = self._imgl_ptrs(self.grads, k, m0)
                wk, _ = self._imgl_ptrs(self.params, k, m0)
                check(lib.cl_wide_image_wgrad(ptr(dz), ldw, hs[l][0], hs[l][1], ptr(seg), seg.numel() - 1, n, w, gw, gb, sf, st), "cl_wide_image_wgrad")
                if w > 128:
                    tl = obs.wide_tiles[a]
                    check(lib.cl_wide_image_dgrad_tiles(ptr(dz), ldw, wk, ptr(seg), ptr(tl), tl.numel() // 2, w, hs[l][0], hs[l][1], leak, ptr(dzn), ldw, sf, st),
                          "cl_wide_image_dgrad_tiles")
                else:
                    check(lib.cl_wide_image_dgrad(ptr(dz), ldw, wk, ptr(seg), seg.numel() - 1, n, w, hs[l][0], hs[l][1], leak, ptr(dzn), ldw, sf, st),
                          "cl_wide_image_dgrad")
                dz, dzn = dzn, dz
            pre = self._wide_pre()
            for l in range(self.L - (2 if headb else 1), -1, -1):
                ow, ob, fan_in = layers[l]
                if pre and l == 1:
                    # the layer's input h_0 is recomputed from the metadata: as the weight gradient's operand and as the dgrad's mask
                    ow0, ob0, d0 = layers[0]
                    x0, ld0 = hs[0]
                    check(lib.cl_wide_dense_wgrad_pre(ptr(dz), ldw, x0, ld0, d0, pbase + 4 * ow0, pbase + 4 * ob0, leak, n, w, fan_in, ptr(W["wpart"]),
                                                      nsplit, sf, st), "cl_wide_dense_wgrad_pre")
                    check(lib.cl_reduce_partials(ptr(W["wpart"]), nsplit, w * fan_in + w, gbase + 4 * ow, sf, st), "cl_reduce_partials")
                    # layer 1's dgrad with layer 0's weight gradient taken where dZ_0 is produced (it is never stored) ...
                    rc = -2 if not self.FUSE_WG0 else \
                        lib.cl_wide_dense_dgrad_pre_wgrad0(ptr(dz), ldw, pbase + 4 * ow, n, w, fan_in, x0, ld0, d0, pbase + 4 * ow0, pbase + 4 * ob0, leak,
                                                           ptr(W["wpart"]), sf, st)
                    if rc != -2:
                        check(rc, "cl_wide_dense_dgrad_pre_wgrad0")
                        check(lib.cl_reduce_partials(ptr(W["wpart"]), int(lib.cl_wide_dgrad_wgrad0_parts(n)), w * d0 + w, gbase + 4 * ow0, sf, st),
                              "cl_reduce_partials")
                        break
                    # ... or, outside that kernel's envelope, the two launches
                    check(lib.cl_wide_dense_dgrad_pre(ptr(dz), ldw, pbase + 4 * ow, n, w, fan_in, x0, ld0, d0, pbase + 4 * ow0, pbase + 4 * ob0, leak,
                                                      ptr(dzn), ldw, sf, st), "cl_wide_dense_dgrad_pre")
                    dz, dzn = dzn, dz
                    continue
                check(lib.cl_wide_dense_wgrad(ptr(dz), ldw, hs[l][0], hs[l][1], n, w, fan_in, ptr(W["wpart"]), nsplit, sf, st), "cl_wide_dense_wgrad")
                check(lib.cl_reduce_partials(ptr(W["wpart"]), nsplit, w * fan_in + w, gbase + 4 * ow, sf, st), "cl_reduce_partials")
                if l > 0:
                    check(lib.cl_wide_dense_dgrad(ptr(dz), ldw, pbase + 4 * ow, n, w, fan_in, hs[l][0], hs[l][1], leak, ptr(dzn), ldw, sf, st),
                          "cl_wide_dense_dgrad")
                    dz, dzn = dzn, dz

