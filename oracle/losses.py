"""ORACLE (test infrastructure): the three target-side losses of one adaptation step.

Parity status: PINNED by fixture G3 (values and gradients from the reference functions).

* ce_hard   framework/utils/func.py:35-42 -> framework/utils/loss.py:16-45
* rce_hard  framework/utils/loss.py:88-112 (hard-label branch)
* mrkld     framework/domain_adaptation/methods/prototypes.py:29-39 ("MRKLD")
"""
import torch
import torch.nn.functional as F

LOG_CLAMP = 1e-4  # loss.py:104-106: one-hot clamped to [1e-4, 1] before the log


def ce_hard(logits, target):
    """Mean cross-entropy over pixels whose target is neither negative nor 255.
    An all-ignored target gives NaN (mean over an empty set), as the reference does."""
    target = target.long()
    n, c, h, w = logits.shape
    keep = (target >= 0) & (target != 255)
    rows = logits.permute(0, 2, 3, 1)[keep]
    return F.cross_entropy(rows.reshape(-1, c), target[keep])


def rce_hard(logits, target):
    """Reverse cross-entropy: -sum_px mask * sum_c p_c * log(clamp(onehot_c)) / (sum mask + 1e-6)."""
    p = logits.softmax(dim=1)
    c = p.shape[1]
    t = target.long().clone()
    mask = (t != 255).float()
    t[t == 255] = c
    onehot = F.one_hot(t, c + 1).float().permute(0, 3, 1, 2)[:, :-1]
    log_oh = torch.log(torch.clamp(onehot, min=LOG_CLAMP, max=1.0))
    return -((p * log_oh).sum(dim=1) * mask).sum() / (mask.sum() + 1e-6)


def mrkld(logits):
    """-mean over every element of log_softmax(logits)."""
    return -F.log_softmax(logits, dim=1).sum() / logits.numel()


def target_loss(logits, pseudo, ce_w=0.1, rce_w=1.0, reg_w=0.1):
    """Total target loss of prototypes.py:313-328 for hard labels with the
    hybrid_switch.yml weights (RCE_ALPHA, RCE_BETA, REGULARIZER_WEIGHT)."""
    parts = {"ce_loss": ce_hard(logits, pseudo), "rce_loss": rce_hard(logits, pseudo),
             "regularization_loss": mrkld(logits)}
    sym = ce_w * parts["ce_loss"] + rce_w * parts["rce_loss"]
    total = sym + reg_w * parts["regularization_loss"]
    # the reference aliases total_loss = sym_loss and then adds in place, so the
    # logged "sym_loss" equals the total
    parts["sym_loss"] = total
    parts["Total target loss"] = total
    return parts
