"""Drop-in for ``framework/domain_adaptation/methods/prototype_handler.py``: prototype state,
feature<->prototype distance, pseudo-label assignment and prototype EMA on HIP kernels.

Same constructor, attributes (``prototypes`` / ``squared_mean`` / ``counter`` -- the int 0
before initialisation), pickle 3-tuple format (:37-47) and methods.  One kernel pass per
(feat, prior) computes the hard labels AND the soft map AND the monitor means, so the
reference's second ``pseudo_labels(..., soft=True)`` call (prototypes_hybrid_switch.py:93)
is served from that pass instead of recomputing all distances.
"""
import os
import pickle

import torch

from onda_amd import ops
from onda_amd._lib import call, query

_p = ops._p


def _rows(m):
    """(tensor, ld, N, C) of a feature / prior map in pixel-major order without copying when
    it already is (NCHW-shaped view of an NHWC buffer, or a 2-D [N,C] matrix)."""
    if m.dim() == 2:
        m = m if m.stride(1) == 1 else m.contiguous()
        return m, m.stride(0), m.shape[0], m.shape[1]
    B, C, H, W = m.shape
    ld = m.stride(3)
    if m.stride(1) == 1 and m.stride(2) == W * ld and (B == 1 or m.stride(0) == H * W * ld) and ld >= C:
        return m, ld, B * H * W, C
    buf = m.permute(0, 2, 3, 1).contiguous()
    return buf, C, B * H * W, C


class prototype_handler:
    def __init__(self, ma_lambda=0.9999, tau=1, thresh=0, distance_metric="euclidean",
                 confidence_regularization_threshold=1):
        self.prototypes = 0  # classes x features once initialised
        self.squared_mean = 0
        self.counter = 0
        self.ma_lambda = ma_lambda
        self.tau = tau
        self.thresh = thresh
        if distance_metric not in ("euclidean", "mahalanobis"):
            raise ValueError("unexpected value for attribute distance_metric")
        self.distance_metric = distance_metric
        self.confidence_regularization_threshold = (
            1 if isinstance(confidence_regularization_threshold, dict) else confidence_regularization_threshold)
        self._cache_key = None
        self._cache = None
        self._cache_args = (None, None)
        self._stats_host = None

    # ---- persistence (same 3-tuple pickle as the reference) ---------------------------------
    def save(self, loc="prototypes.pickle"):
        pickle.dump((self.prototypes, self.squared_mean, self.counter), open(loc, "wb"))

    def load(self, loc="prototypes.pickle"):
        if os.path.exists(loc):
            self.prototypes, self.squared_mean, self.counter = pickle.load(open(loc, "rb"))
            print("Prototypes loaded!")
            return True
        return False

    def to(self, device):
        if isinstance(self.prototypes, torch.Tensor):
            self.prototypes = self.prototypes.to(device).float().contiguous()
            self.squared_mean = self.squared_mean.to(device).float().contiguous()
            self.counter = self.counter.to(device).float().contiguous()
        return self

    # ---- small derived quantities -----------------------------------------------------------
    def prototype_var(self):
        return torch.sqrt(self.squared_mean - self.prototypes ** 2)

    def global_var(self):
        """Per-channel std shared by all classes (reference :53-60), one tiny kernel."""
        K, C = self.prototypes.shape
        sigma = torch.empty(C, device=self.prototypes.device, dtype=torch.float32)
        call("onda_proto_sigma", _p(self.prototypes), _p(self.squared_mean), _p(self.counter), _p(sigma), K, C,
             ops._stream())
        return sigma

    def transform(self, matrix):
        if matrix.dim() == 2:
            return matrix
        return matrix.permute(0, 2, 3, 1).reshape(-1, matrix.size(1))

    # ---- class statistics ------------------------------------------------------------------------
    def _class_sums(self, feat, out, classes=None):
        """sums[2][K][C], counts[K] of feat and feat^2 under argmax(out) (reference :76-86).
        `out` may be logits / one-hot ([N,K] or NCHW); `classes` (i32[N], values outside
        [0,K) = row dropped) short-cuts the argmax when the caller already has the index."""
        frows, ldf, N, C = _rows(feat)
        if classes is not None:
            cls = classes.reshape(-1).to(device=frows.device, dtype=torch.int32).contiguous()
            K = out if isinstance(out, int) else out.shape[1]
        elif out.dim() == 2:
            cls = out.argmax(dim=1).to(torch.int32)
            K = out.shape[1]
        else:
            _, _, cls = ops.softmax_stats(out, want_argmax=True)
            K = out.shape[1]
        dev = frows.device
        sums = torch.empty(2, K, C, device=dev, dtype=torch.float32)
        counts = torch.empty(K, device=dev, dtype=torch.float32)
        ws = torch.empty(query("onda_proto_sums_ws", N, C, K), device=dev, dtype=torch.float32)
        call("onda_proto_class_sums", _p(frows), ldf, _p(cls), _p(sums), _p(counts), _p(ws), N, C, K, ops._stream())
        return sums, counts, K, C

    def append(self, feat, out, classes=None):
        sums, counts, K, C = self._class_sums(feat, out, classes)
        self._append_sums(sums, counts, K, C)

    def append_from_statistics(self, flat, K, C):
        """`append` from the flat [sum feat | sum feat^2 | count] buffer of class_statistics (the multi-GPU path
        sums it over ranks first, so that every rank starts from the same prototypes)."""
        self._append_sums(flat[: 2 * K * C].reshape(2, K, C), flat[2 * K * C:], K, C)

    def _append_sums(self, sums, counts, K, C):
        if type(self.prototypes) == int:
            self.prototypes = torch.zeros(K, C, device=sums.device)
            self.squared_mean = torch.zeros(K, C, device=sums.device)
            self.counter = torch.zeros(K, device=sums.device)
        call("onda_proto_append", _p(self.prototypes), _p(self.squared_mean), _p(self.counter), _p(sums), _p(counts),
             K, C, ops._stream())
        self._touch()

    def class_statistics(self, feat, out, classes=None):
        """The raw batch statistics as ONE flat buffer [sum feat | sum feat^2 | count] -- the
        unit the multi-GPU path all-reduces before the blend."""
        sums, counts, K, C = self._class_sums(feat, out, classes)
        return torch.cat([sums.reshape(-1), counts]), K, C

    def ma_from_statistics(self, flat, K, C):
        sums, counts = flat[: 2 * K * C], flat[2 * K * C:]
        call("onda_proto_ema", _p(self.prototypes), _p(self.squared_mean), _p(sums), _p(counts), float(self.ma_lambda),
             K, C, ops._stream())
        self._touch()

    def ma(self, feat, out):
        flat, K, C = self.class_statistics(feat, out)
        self.ma_from_statistics(flat, K, C)

    def onehot(self, matrix):
        """One-hot rows of the row-wise argmax (reference :83-86)."""
        return torch.zeros_like(matrix, dtype=torch.float32).scatter_(1, matrix.argmax(dim=1, keepdim=True), 1.0)

    def get_proto_array(self, feat, out):
        """(per-class feature sums [K,C], per-class pixel counts [K]) under argmax(out) (reference :76-81) -- the
        class-sum kernel's first output; `ma` / `append` use both moments at once."""
        sums, counts, _, _ = self._class_sums(feat, out)
        return sums[0], counts

    # ---- distance matrices by themselves (reference :111-138; pseudo_labels computes them inside its one pass) -------
    def _distances(self, feat, maha):
        frows, ldf, N, C = _rows(feat)
        K = self.prototypes.shape[0]
        dist = torch.empty(N, K, device=frows.device, dtype=torch.float32)
        call("onda_proto_distances", _p(frows), ldf, _p(self.prototypes), _p(self.global_var()) if maha else None, int(maha),
             _p(dist), N, C, K, ops._stream())
        return dist

    def mahalanobis_distance(self, feat):
        return self._distances(feat, True)

    def distance(self, feat):
        return self._distances(feat, False)

    @property
    def distance_measure(self):
        return self.mahalanobis_distance if self.distance_metric == "mahalanobis" else self.distance

    def _touch(self):
        self._cache_key = None
        for t in (self.prototypes, self.squared_mean, self.counter):
            torch.autograd.graph.increment_version(t)

    # ---- distances / pseudo-labels ------------------------------------------------------------------
    def _assign(self, feat, prior):
        # the cache holds the argument tensors themselves: identity (not address) decides a hit
        key = (feat._version, None if prior is None else prior._version, float(self.tau), float(self.thresh),
               self.prototypes._version, self.distance_metric)
        if self._cache_key == key and self._cache_args[0] is feat and self._cache_args[1] is prior:
            return self._cache
        frows, ldf, N, C = _rows(feat)
        K = self.prototypes.shape[0]
        prows, ldp = None, 0
        if prior is not None:
            if prior.device != frows.device:
                print(f"vetors not in the same device, feat: {feat.device}, prior: {prior.device}")
            prows, ldp, _, _ = _rows(prior)
        maha = self.distance_metric == "mahalanobis"
        sigma = self.global_var() if maha else None
        dev = frows.device
        labels = torch.empty(N, 1, device=dev, dtype=torch.int64)
        soft = torch.empty(N, K, device=dev, dtype=torch.float32)
        result = torch.empty(3, device=dev, dtype=torch.float32)
        ws = torch.empty(3 * query("onda_proto_assign_blocks", N), device=dev, dtype=torch.float32)
        call("onda_proto_assign", _p(frows), ldf, _p(prows), ldp, _p(self.prototypes), _p(sigma), int(maha),
             float(self.tau), float(self.thresh), _p(labels), _p(soft), _p(result), _p(ws), N, C, K, ops._stream())
        self._cache_key, self._cache, self._cache_args = key, (labels, soft, result), (feat, prior)
        self._stats_host = None
        return self._cache

    def assign_stats(self, feat, prior, reduce=None):
        """(labels i64[N,1], soft f32[N,K], [mean max softmax(-D/tau), mean max posterior,
        mean max prior] read back to the host once) -- everything one pass produces.  `reduce` (the
        multi-GPU path: mean over ranks) is applied to the three device scalars before the read-back, so that
        every rank's monitor -- and the tau bump it drives -- sees the same numbers."""
        out = self._assign(feat, prior)
        if self._stats_host is None:
            self._stats_host = (reduce(out[2].clone()) if reduce is not None else out[2]).tolist()
        return out[0], out[1], self._stats_host

    def pseudo_labels(self, feat, prior=None, soft=False, confidence_monitor=None):
        labels, soft_map, stats = self._assign(feat, prior)
        if confidence_monitor is not None and not confidence_monitor.freeze:
            confidence_monitor.add({"prototypes": self._stats_host[0] if self._stats_host is not None else stats[0]})
            if confidence_monitor.avg("prototypes") > self.confidence_regularization_threshold:
                self.tau += 0.001
                confidence_monitor.add({"tau": self.tau})
        return soft_map if soft else labels
