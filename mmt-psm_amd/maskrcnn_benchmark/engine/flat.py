"""Flat parameter storage: all parameters of a model live in ONE fp32 device buffer
[trainable weights | trainable biases | frozen], each nn.Parameter being a view (conv weights keep their
channels_last strides).  Gradients and SGD momentum mirror the layout.  This is what turns the reference's
132-tensor EMA loop (engine/MTtrainer.py:277-281), its 121-group SGD (solver/build.py:18) and the (new) data
parallel gradient all-reduce into single launches / single collectives over contiguous memory."""
import torch


def _is_bias(name):
    return "bias" in name  # solver/build.py:14


class FlatParams(object):
    def __init__(self, model):
        named = list(model.named_parameters())
        tw = [(n, p) for n, p in named if p.requires_grad and not _is_bias(n)]
        tb = [(n, p) for n, p in named if p.requires_grad and _is_bias(n)]
        fz = [(n, p) for n, p in named if not p.requires_grad]
        dev = named[0][1].device

        def pad4(n):
            return (n + 7) // 8 * 8  # 32-byte aligned views: float4 kernels, and 16-byte aligned bf16 plane slices

        self.index = {}
        off = 0
        for group in (tw, tb, fz):
            for n, p in group:
                self.index[n] = (off, p.numel())
                off += pad4(p.numel())  # 16-byte aligned views for the float4 kernels
            off = pad4(off)
            if group is tw:
                self.n_weights = off
            elif group is tb:
                self.n_biases = off - self.n_weights
        self.n_trainable = self.n_weights + self.n_biases
        self.total = off
        self.data = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_trainable, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(self.n_trainable, dtype=torch.float32, device=dev)
        # names of the parameters that received a gradient since the last zero_grad: fused backward nodes report their direct
        # writes (layers/fused.py::_touch, looked up by slot address), autograd deliveries are seen by a hook.  FlatSGD
        # updates only these, which is torch.optim.SGD's `if p.grad is None: continue` (solver/build.py)
        self.touched = set()
        import weakref
        from .. import _hip as H
        me = weakref.ref(self)
        for n, p in named:
            o, k = self.index[n]
            view = self._view_like(self.data[o:o + k], p)
            view.copy_(p.data)
            p.data = view
            if p.requires_grad:
                p.grad = self._view_like(self.grad[o:o + k], p)
                # fused backward nodes accumulate weight gradients straight into this slot (layers/fused.py)
                p._flat_grad = p.grad
                H.GRAD_SLOTS[p.grad.data_ptr()] = (me, n)
                p.register_post_accumulate_grad_hook(lambda q, n=n, me=me: me() is not None and me().touched.add(n))
        self.planes = None
        self.planes16 = self.stat16 = None   # fp16 two-term planes of the same matrices + (max, scale) per matrix
        self.f16_gen = -1
        self.bf16_gen = -1                   # generation the bf16 planes (forward + data-gradient form) were packed for
        self._lazy_bf16 = False
        self.views16 = {}                    # weight address -> (its slice of planes16, its scale in stat16)
        self.plane_versions = {}
        self.plane_epoch = -1
        self.plane_gen = 0
        self._named = named
        self.refresh_planes()

    def refresh_planes(self):
        """split-bf16 conv modes (mmt_set_conv_precision != 0): re-pack every weight matrix of the parameter buffer into
        its three bf16 planes (include/mmtpsm.h: mmt_pack_weights) -- ONE launch over a device descriptor table; called
        after everything that rewrites parameters through raw pointers (SGD, EMA, teacher initialisation).
        Convolutions look their weight up by address in _hip.PLANES."""
        from .. import _hip as H
        self.plane_gen += 1  # the parameters changed: anything derived from them (H.FLIPPED) is stale
        if not self.data.is_cuda or H.get_conv_precision() == 0:
            return
        if self.planes is None:
            import struct
            import weakref
            descs, unit_desc, off, unit0 = [], [], 0, 0
            for n, p in self._named:
                if p.dim() < 2 or p.shape[0] <= 32 or (p.numel() // p.shape[0]) % 16:
                    continue  # such layers always run on the fp32 kernel
                cout, k = p.shape[0], p.numel() // p.shape[0]
                o, _ = self.index[n]
                elems = H.packed_elems(cout, k)
                descs.append(struct.pack("qqiiii", o, off, cout, k, unit0, 0))
                unit_desc += [len(descs) - 1] * (elems // 512)
                H.PLANES[p.data_ptr()] = (weakref.ref(self), off, p.numel(), len(descs) - 1)
                off += elems
                unit0 += elems // 512
            dev = self.data.device
            self.planes = torch.empty((3, max(off, 8)), dtype=torch.bfloat16, device=dev)
            H.LAYOUT_EPOCH[0] += 1   # (recorded launch plans hold plane addresses)
            self._pack_descs = torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev)
            self._pack_units = torch.tensor(unit_desc, dtype=torch.int32, device=dev)
            self._n_units = unit0
            self._n_descs = len(descs)
            self.planes16 = self.stat16 = None
            self.views16 = {}
        # on the fp16 split (the default) next to nothing reads the bf16 planes -- only a site that fell back, or a shape the
        # fp16 kernels do not take: they are packed on first use of a generation (`ensure_bf16`) instead of after every step
        self._lazy_bf16 = bool(H.F16X2 and H.get_conv_precision() == 3 and self._n_descs)
        if not self._lazy_bf16:
            H.pack_weights(self.data, self.planes, self._pack_descs, self._pack_units, self._n_units)
            self.bf16_gen = self.plane_gen
        if H.F16X2 and H.get_conv_precision() == 3 and self._n_descs:
            # the default arithmetic's form of the same matrices: two fp16 planes of w * s, s per matrix from its maximum
            # (reduction launch + packing launch over the same tables); stat16[d] = (max |w_d|, s_d)
            if self.planes16 is None:
                H.LAYOUT_EPOCH[0] += 1
                self.planes16 = torch.empty((2, self.planes.shape[1]), dtype=torch.float16, device=self.data.device)
                self.stat16 = torch.zeros((self._n_descs, 2), dtype=torch.float32, device=self.data.device)
            H._check(H.lib().mmt_pack_weights_f16(self.data.data_ptr(), self.planes16.data_ptr(), self.planes16.stride(0),
                                                  self._pack_descs.data_ptr(), self._pack_units.data_ptr(), self._n_units,
                                                  self._n_descs, self.stat16.data_ptr(), H._stream()), "mmt_pack_weights_f16")
            self.f16_gen = self.plane_gen
        self.plane_versions = {p.data_ptr(): p._version for _, p in self._named if p.dim() >= 2}
        self.plane_epoch = H.PLANES_EPOCH
        self._repack_flipped()

    def register_flipped(self, w, scale, planes, dims):
        """a data-gradient weight (w, BN scale) whose packed planes `planes` should follow the parameters from now on"""
        ent = self.__dict__.setdefault("_flip_entries", {})
        if w.data_ptr() not in ent or ent[w.data_ptr()][1] is not scale:
            ent[w.data_ptr()] = (w, scale, planes, dims)
            self._flip_table = None

    def register_flipped16(self, w, scale, planes, dims):
        """the same for the fp16 two-term planes of a data-gradient weight (default arithmetic of mode 3)"""
        ent = self.__dict__.setdefault("_flip16_entries", {})
        if w.data_ptr() not in ent or ent[w.data_ptr()][1] is not scale:
            ent[w.data_ptr()] = (w, scale, planes, dims)
            self._flip16_table = None

    def flipped16(self, w, scale):
        """(planes, scale view) of a registered data-gradient weight packed for THIS parameter generation, or None"""
        t = self.__dict__.get("_flip16_table")
        if t is None or self.__dict__.get("_flip16_gen") != self.plane_gen:
            return None
        e = self._flip16_entries.get(w.data_ptr())
        # (the address lies inside this buffer, which outlives the entry: the same address with the same dimensions IS the same
        # matrix -- a Linear layer hands its weight over as a fresh 4-D view every call, identity of the object would never match)
        if e is None or e[1] is not scale or (w.shape[0], w.shape[2], w.shape[3], w.shape[1]) != tuple(e[3]):
            return None
        v = t[5].get(w.data_ptr())
        if v is None:
            v = t[5][w.data_ptr()] = (e[2], t[4][t[3][w.data_ptr()], 1:2])
        return v

    def _repack_flipped16(self):
        from .. import _hip as H
        ent = self.__dict__.get("_flip16_entries")
        if not ent or H.get_conv_precision() != 3 or not H.F16X2:
            return
        if self.__dict__.get("_flip16_table") is None:
            import struct
            descs, unit_desc, unit0, idx = [], [], 0, {}
            for w, scale, planes, (Cout, KH, KW, Cin) in ent.values():
                units = planes.shape[1] // 512
                idx[w.data_ptr()] = len(descs)
                descs.append(struct.pack("qqqqiiiiii", w.data_ptr(), 0 if scale is None else scale.data_ptr(), planes.data_ptr(),
                                         planes.stride(0), Cout, KH, KW, Cin, unit0, 0))
                unit_desc += [len(descs) - 1] * units
                unit0 += units
            dev = self.data.device
            self._flip16_table = (torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev),
                                  torch.tensor(unit_desc, dtype=torch.int32, device=dev), unit0, idx,
                                  torch.zeros((len(descs), 2), dtype=torch.float32, device=dev), {})
        d, u, n, idx, stat, _ = self._flip16_table
        H._check(H.lib().mmt_pack_weights_flipped_f16(d.data_ptr(), u.data_ptr(), n, len(idx), stat.data_ptr(), H._stream()),
                 "mmt_pack_weights_flipped_f16")
        self._flip16_gen = self.plane_gen

    def _repack_flipped(self):
        from .. import _hip as H
        self._repack_flipped16()
        ent = self.__dict__.get("_flip_entries")
        if not ent or H.get_conv_precision() == 0:
            return
        d, u, n = self._flip_table_now()
        if not self._lazy_bf16:
            H._check(H.lib().mmt_pack_weights_flipped(d.data_ptr(), u.data_ptr(), n, H._stream()), "mmt_pack_weights_flipped")
        for w, scale, planes, _ in ent.values():
            H.FLIPPED[w.data_ptr()] = ((id(self), self.plane_gen, H._p(scale), None if scale is None else scale._version), planes)

    def _flip_table_now(self):
        """descriptor table over the registered data-gradient weights (bf16 planes), rebuilt when an entry was registered
        since it was made"""
        if self.__dict__.get("_flip_table") is None:
            import struct
            descs, unit_desc, unit0 = [], [], 0
            for w, scale, planes, (Cout, KH, KW, Cin) in self._flip_entries.values():
                units = planes.shape[1] // 512
                descs.append(struct.pack("qqqqiiiiii", w.data_ptr(), 0 if scale is None else scale.data_ptr(), planes.data_ptr(),
                                         planes.stride(0), Cout, KH, KW, Cin, unit0, 0))
                unit_desc += [len(descs) - 1] * units
                unit0 += units
            dev = self.data.device
            self._flip_table = (torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev),
                                torch.tensor(unit_desc, dtype=torch.int32, device=dev), unit0)
        return self._flip_table

    def ensure_bf16(self):
        """the bf16 planes of this parameter generation, packed now if they were deferred (called right before a launch that
        reads them, on that launch's stream: all consumers of a model's planes sit on one stream)"""
        if self.bf16_gen == self.plane_gen or self.planes is None:
            return
        from .. import _hip as H
        H.pack_weights(self.data, self.planes, self._pack_descs, self._pack_units, self._n_units)
        if self.__dict__.get("_flip_entries"):
            # a weight registered since the last bulk pack (register_flipped drops the table) must not leave the OTHER entries
            # un-packed: their H.FLIPPED keys already carry this generation
            t = self._flip_table_now()
            H._check(H.lib().mmt_pack_weights_flipped(t[0].data_ptr(), t[1].data_ptr(), t[2], H._stream()), "mmt_pack_weights_flipped")
        self.bf16_gen = self.plane_gen

    def active_ranges(self, names, lo, hi):
        """merged [a, b) element ranges inside [lo, hi) of the flat buffer covered by the parameters in `names`
        (padding between neighbours included, so adjacent parameters merge into one range)"""
        spans = sorted(self.index[n] for n in names if lo <= self.index[n][0] < hi)
        order = sorted(o for o, _ in self.index.values() if lo <= o < hi) + [hi]
        nxt = {o: order[i + 1] for i, o in enumerate(order[:-1])}
        out = []
        for o, _ in spans:
            b = nxt[o]
            if out and out[-1][1] == o:
                out[-1][1] = b
            else:
                out.append([o, b])
        return [(a, b) for a, b in out]

    @staticmethod
    def _view_like(flat, p):
        if p.dim() == 4:  # memory is [O][H][W][I] (channels_last) for every 4-D weight on the path
            o, i, h, w = p.shape
            return flat.view(o, h, w, i).permute(0, 3, 1, 2)
        return flat.view(p.shape)


def flatten_model(model):
    if getattr(model, "_flat", None) is None:
        model._flat = FlatParams(model)
    return model._flat
