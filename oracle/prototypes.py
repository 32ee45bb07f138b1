"""ORACLE (test infrastructure): CPU restatement of the reference's prototype
state, feature<->prototype distance, pseudo-label assignment and prototype EMA
(``framework/domain_adaptation/methods/prototype_handler.py``).

Parity status: PINNED by fixture G4 (``tests/golden/make_golden.py`` runs the
reference class itself and stores its outputs).

All functions take / return plain tensors; `state` is the pickle 3-tuple
``(prototypes f32[K,C], squared_mean f32[K,C], counter f32[K])``
(prototype_handler.py:37-47).
"""
import torch


def to_rows(m):
    """NCHW -> pixel-major rows [B*h*w, C] (prototype_handler.py:105-109)."""
    if m.dim() == 2:
        return m
    return m.permute(0, 2, 3, 1).reshape(-1, m.shape[1])


def global_std(state):
    """Per-channel std shared by all classes (prototype_handler.py:53-60)."""
    proto, sqmean, counter = state
    w = counter / counter.sum()
    g_sq = (sqmean.T * w).T.sum(0)
    g_mean = (proto.T * w).T.sum(0)
    return torch.sqrt(g_sq - g_mean ** 2)


def distances(feat, state, metric="mahalanobis"):
    """D[n,k] minus its row minimum (prototype_handler.py:111-138).  The class loop
    and torch.norm call mirror the reference so the fp32 values agree with it."""
    proto = state[0]
    rows = to_rows(feat)
    d = torch.ones(rows.shape[0], proto.shape[0])
    if metric == "mahalanobis":
        sigma = global_std(state)
        for k in range(proto.shape[0]):
            d[:, k] = torch.norm((rows - proto[k]) / sigma, 2, dim=1)
    elif metric == "euclidean":
        for k in range(proto.shape[0]):
            d[:, k] = torch.norm(rows - proto[k], 2, dim=1)
    else:
        raise ValueError("unexpected value for attribute distance_metric")
    return (d.T - d.min(dim=1)[0]).T


def assign(feat, prior, state, tau=1.0, thresh=0.0, metric="mahalanobis"):
    """prototype_handler.pseudo_labels (:140-166), both outputs at once.

    Returns (labels i64[N,1] with 255 = below threshold, soft f32[N,K],
    proto_conf = mean over pixels of max_k softmax(-D/tau) -- the value the
    reference feeds its confidence monitor at :150)."""
    d = distances(feat, state, metric)
    prop = (-d / tau).softmax(dim=1)
    proto_conf = prop.max(dim=1)[0].mean()
    if prior is not None:
        prop = prop * to_rows(prior)
    prop = prop / prop.sum(dim=1, keepdim=True)
    mprop, labels = prop.max(dim=1, keepdim=True)
    labels = labels.clone()
    labels[mprop < thresh] = 255
    return labels, prop, proto_conf


def class_sums(feat, out):
    """Per-class sums under the one-hot of argmax(out) (prototype_handler.py:76-86):
    (S f32[K,C], n f32[K])."""
    rows, o = to_rows(feat), to_rows(out)
    onehot = torch.zeros_like(o).float().scatter(1, o.argmax(dim=1, keepdim=True), 1)
    return onehot.T @ rows, onehot.sum(0)


def ema_update(state, feat, out, lam):
    """prototype_handler.ma (:88-99): classes present in the batch move towards the
    batch mean with weight (1 - lam); absent classes are untouched."""
    proto, sqmean, counter = state
    s, n = class_sums(feat, out)
    s2, _ = class_sums(feat ** 2, out)
    keep = lam ** (n > 0).float()
    n_safe = torch.where(n > 0, n, torch.ones_like(n))
    proto = (proto.T * keep).T + ((1 - keep) * (s.T / n_safe)).T
    sqmean = (sqmean.T * keep).T + ((1 - keep) * (s2.T / n_safe)).T
    return proto, sqmean, counter


def running_append(state, feat, out):
    """prototype_handler.append (:62-74): running per-class mean / mean of squares.
    `state` may be None before the first batch."""
    s, n = class_sums(feat, out)
    s2, _ = class_sums(feat ** 2, out)
    if state is None:
        proto, sqmean, counter = torch.zeros_like(s), torch.zeros_like(s), torch.zeros_like(n)
    else:
        proto, sqmean, counter = (t.clone() for t in state)
    counter = counter + n
    denom = torch.where(counter > 0, counter, torch.ones_like(counter))
    proto = proto + ((s - (proto.T * n).T).T / denom).T
    sqmean = sqmean + ((s2 - (sqmean.T * n).T).T / denom).T
    return proto, sqmean, counter
