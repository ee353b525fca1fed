"""AdamW on the HIP kernel: one launch per step for all parameters, bf16 shadows refreshed in the same pass.

Stands in for ``torch.optim.AdamW`` as the swin configs build it (``configs/swin/*_coco.py:64-67``: lr 1e-4,
betas (0.9, 0.999), weight_decay 0.05 with 0 for norm / position-bias parameters via ``paramwise_cfg``) plus the
master->half copy of apex O1 (``mmdet/apis/train.py:82-89``).  Gradients are read from ``p.grad`` -- with
``ddp.BucketedGradReducer`` these are stable views of the flat all-reduce buckets."""
import ctypes
import struct

import torch

from ._lib import half_dtype as _H

from . import _lib, mixed
from ._lib import SwinHipError, call


class FusedAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [dict(params=groups)]
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g['params'] = [p for p in g['params']]
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            self.param_groups.append(g)
        if len(self.param_groups) > 8:
            raise SwinHipError("FusedAdamW: at most 8 parameter groups")
        if len({(tuple(g['betas']), g['eps']) for g in self.param_groups}) != 1:
            raise SwinHipError("FusedAdamW: betas / eps must be shared by all groups")
        self.state = {}
        self._flat_params = [p for g in self.param_groups for p in g['params']]
        self._partial_done = set()
        self.step_count = 0
        self._tables = None
        self._state = None                  # device-resident per-step scalars (32 floats, include/swin_hip.h: swin_adamw_step_dev)
        self._prepared = False

    # -- tables: built on the first step (gradients must exist), rebuilt if any pointer moved --------------------
    def _signature(self):
        """Everything the device-resident table depends on: each parameter's storage, its gradient's storage and its shadow.
        Flat list passes instead of a tuple per parameter: this check runs every step."""
        ps = self._flat_params
        sh = mixed._SHADOW
        return ([p.data_ptr() for p in ps], [0 if p.grad is None else p.grad.data_ptr() for p in ps],
                [id(sh.get(id(p))) if id(p) in sh else id(mixed.shadow_of(p)) for p in ps])

    def _build(self):
        chunk = _lib.lib().swin_adamw_chunk_elems()
        segs, chunks, rows = [], [], {}
        dev = None
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                if not p.requires_grad or p.grad is None:
                    continue
                # the kernel pairs parameter, gradient, moments and shadow by memory offset: all five must be dense with ONE
                # layout (contiguous, or channels-last for the resident 3x3 conv weights -- mixed.khwc_resident_)
                dense = p.is_contiguous() or mixed.is_khwc(p)
                if not (p.is_cuda and p.dtype == torch.float32 and dense and p.grad.dtype == torch.float32
                        and p.grad.stride() == p.stride()):
                    raise SwinHipError("FusedAdamW: dense fp32 GPU parameters with gradients of the same memory layout only")
                dev = p.device
                st = self.state.setdefault(p, {})
                if 'exp_avg' not in st:
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st['exp_avg'].stride() != p.stride() or st['exp_avg_sq'].stride() != p.stride():
                    raise SwinHipError("FusedAdamW: optimizer state laid out differently from its parameter")
                sh = mixed.shadow_of(p)
                if sh is not None and not (sh.dtype == _H() and sh.stride() == p.stride() and sh.numel() == p.numel()):
                    raise SwinHipError("FusedAdamW: shadows must be bf16 copies with the parameter's memory layout")
                n = p.numel()
                si = len(segs)
                segs.append(struct.pack("<QQQQQqii", p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(),
                                        st['exp_avg_sq'].data_ptr(), sh.data_ptr() if sh is not None else 0, n, gi, 0))
                rows[id(p)] = (len(chunks), (n + chunk - 1) // chunk)
                chunks += [(si, c) for c in range((n + chunk - 1) // chunk)]
        self._rows, self._chunks_host, self._subsets = rows, chunks, {}
        if not segs:
            self._tables = (None, None, 0)
            self._sig = self._signature()
            return
        seg_t = torch.frombuffer(bytearray(b"".join(segs)), dtype=torch.uint8).to(dev)
        ck_t = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self._tables = (seg_t, ck_t, len(chunks))
        self._sig = self._signature()

    def state_tensor(self, device):
        """The 32-float device block the kernel reads its per-step scalars from (lr / weight decay per group, bias corrections,
        gradient scale, skip flag, loss-scale state) -- created on first use: grad_scale 1, skip 0, loss scale 1."""
        if self._state is None or self._state.device != device:
            st = torch.zeros(32, dtype=torch.float32)
            st[18] = 1.0
            st[20] = 1.0
            self._state = st.to(device)
        return self._state

    def _launch_dev(self, stream):
        seg_t, ck_t, n_chunks = self._tables
        g0 = self.param_groups[0]
        b1, b2 = g0['betas']
        call("swin_adamw_step_dev", seg_t.data_ptr(), ck_t.data_ptr(), n_chunks, self.state_tensor(seg_t.device).data_ptr(),
             float(b1), float(b2), float(g0['eps']), stream)

    @torch.no_grad()
    def prepare_step(self):
        """First half of step(): (re)build the pointer tables if anything moved, advance the step count and write this step's
        learning rates / weight decays / bias corrections into the device-resident state (one tiny launch, the values travel as
        kernel arguments).  A training step captured in a hipGraph calls this BEFORE each replay and has apply_step() inside the
        graph: the captured optimizer launch then reads this step's scalars."""
        if self._tables is None or self._sig != self._signature():
            self._build()
        self.step_count += 1
        seg_t = self._tables[0]
        if seg_t is None:
            return
        g0 = self.param_groups[0]
        b1, b2 = g0['betas']
        ng = len(self.param_groups)
        lr = (ctypes.c_float * ng)(*[float(g['lr']) for g in self.param_groups])
        wd = (ctypes.c_float * ng)(*[float(g['weight_decay']) for g in self.param_groups])
        with torch.cuda.device(seg_t.device):
            call("swin_adamw_set_state", self.state_tensor(seg_t.device).data_ptr(), lr, wd, ng, 1.0 - b1 ** self.step_count,
                 1.0 - b2 ** self.step_count, torch.cuda.current_stream().cuda_stream)
        self._prepared = True

    @torch.no_grad()
    def apply_step(self):
        """Second half of step(): the AdamW launch itself (+ the bf16 shadow refresh), every scalar read from device memory."""
        if self._tables is None:
            raise SwinHipError("FusedAdamW.apply_step before prepare_step")
        if self._tables[2]:
            self._launch_dev(torch.cuda.current_stream().cuda_stream)
        self._prepared = False
        mixed.shadows_refreshed()

    def _launch(self, ck_t, n_chunks, step_no, stream):
        seg_t = self._tables[0]
        g0 = self.param_groups[0]
        b1, b2 = g0['betas']
        ng = len(self.param_groups)
        lr = (ctypes.c_float * ng)(*[float(g['lr']) for g in self.param_groups])
        wd = (ctypes.c_float * ng)(*[float(g['weight_decay']) for g in self.param_groups])
        call("swin_adamw_step", seg_t.data_ptr(), ck_t.data_ptr(), n_chunks, lr, wd, ng, float(b1), float(b2), float(g0['eps']),
             1.0 - b1 ** step_no, 1.0 - b2 ** step_no, stream)

    def _subset(self, key, ids):
        """device chunk table of the parameters ``ids`` (cached: the reducer's buckets are the same every step)"""
        t = self._subsets.get(key)
        if t is None:
            rows = []
            for i in ids:
                r = self._rows.get(i)
                if r is not None:
                    rows += self._chunks_host[r[0]:r[0] + r[1]]
            dev = self._tables[0].device
            t = self._subsets[key] = (torch.tensor(rows, dtype=torch.int32).reshape(-1, 2).to(dev), len(rows))
        return t

    @torch.no_grad()
    def step_partial(self, params, stream):
        """This step's update for ``params`` only, launched on ``stream`` (a raw handle) NOW -- for a reducer that knows these
        gradients are final while backward is still running (one process: ddp.BucketedGradReducer.early_step).  The caller orders
        the stream behind everything that still reads the parameters or writes their gradients.  step() then updates the rest.
        Returns False (nothing done) until the tables exist, i.e. during the first step."""
        if self._tables is None or self._tables[2] == 0:
            return False
        ids = tuple(id(p) for p in params)
        ck, n = self._subset(ids, ids)
        if n:
            self._launch(ck, n, self.step_count + 1, stream)
        self._partial_done.update(ids)
        return True

    @torch.no_grad()
    def step(self):
        done = self._partial_done
        if not done:
            if not self._prepared:
                self.prepare_step()
            return self.apply_step()
        if self._tables is None or (not done and self._sig != self._signature()):
            self._build()                       # (with early partial steps in flight the tables are by construction the current ones)
        seg_t, ck_t, n_chunks = self._tables
        self.step_count += 1
        if n_chunks == 0:
            return
        stream = torch.cuda.current_stream().cuda_stream
        if done:
            rest = tuple(i for i in self._rows if i not in done)
            ck, n = self._subset(('rest',) + tuple(sorted(done)), rest)
            if n:
                self._launch(ck, n, self.step_count, stream)
            self._partial_done = set()
            if self._sig != self._signature():      # something moved since the tables were built: rebuild before the next use
                self._tables = None
        else:
            self._launch(ck_t, n_chunks, self.step_count, stream)
        mixed.shadows_refreshed()

    def zero_grad(self, set_to_none=False):
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def state_dict(self):
        """torch.optim.AdamW-compatible layout (state per parameter index: step, exp_avg, exp_avg_sq)."""
        idx, state, groups = 0, {}, []
        for g in self.param_groups:
            ids = []
            for p in g['params']:
                st = self.state.get(p)
                if st:
                    state[idx] = dict(step=torch.tensor(float(self.step_count)), exp_avg=st['exp_avg'], exp_avg_sq=st['exp_avg_sq'])
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != 'params'}, 'params': ids})
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        """Resume (mmcv_custom/runner/epoch_based_runner.py:70-104 restores ``checkpoint['optimizer']``): accepts this
        class's state_dict or one written by torch.optim.AdamW over the same parameter order.  Moments are copied into
        this optimizer's own buffers (the kernel's pointer table stays valid), hyper-parameters of the groups are taken
        from the checkpoint."""
        groups = sd['param_groups']
        if len(groups) != len(self.param_groups) or any(len(a['params']) != len(b['params']) for a, b in zip(groups, self.param_groups)):
            raise ValueError("FusedAdamW.load_state_dict: parameter groups do not match")
        steps = set()
        for g_src, g in zip(groups, self.param_groups):
            for k, v in g_src.items():
                if k != 'params' and k in ('lr', 'betas', 'eps', 'weight_decay', 'initial_lr'):
                    g[k] = tuple(v) if k == 'betas' else v
            for idx, p in zip(g_src['params'], g['params']):
                st_src = sd['state'].get(idx)
                if not st_src:
                    continue
                if tuple(st_src['exp_avg'].shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdamW.load_state_dict: moment shape {tuple(st_src['exp_avg'].shape)} != {tuple(p.shape)}")
                st = self.state.setdefault(p, {})
                for name in ('exp_avg', 'exp_avg_sq'):
                    if name not in st:
                        st[name] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st[name].copy_(st_src[name])
                steps.add(int(float(st_src['step'])))
        if len(steps) > 1:
            raise ValueError("FusedAdamW.load_state_dict: one step count for all parameters (the bias corrections are shared)")
        self.step_count = steps.pop() if steps else 0
