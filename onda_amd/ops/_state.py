"""Mutable switches and constants of the operator layer: ONE place, so that `onda_amd.ops.X = v` (forwarded by the package) and
every submodule see the same value."""
import os

BN_EPS = 1e-5
GN_EPS = 1e-5
GN_GROUPS = 32
HEAD_PAD = 32  # the 19-class head is computed as a 32-wide GEMM (padded rows are zero)
STEM_K = 160   # 7*7*3 = 147 patch values padded to a multiple of 32


# How the convolutions (forward, data gradient, weight gradient) are evaluated:
#   "f16x2" : fp32 operands scaled by a per-tensor power of two and split into two f16 limbs, three
#             products on the f16 MFMA pipe with fp32 accumulation (the accuracy of an fp32 FMA chain at
#             16/3 of the fp32-MFMA rate; csrc/conv_l2.hip; the arithmetic is described in conv_h2.hip) -- the default;
#   "f32"   : v_mfma_f32_32x32x2_f32 (an exact fp32 fmaf chain; csrc/conv.hip) -- the strict-fp32 leg of bench.py and the
#             yardstick of test_f16x2_steps_track_the_exact_f32_steps; also what a conv whose weight the f16x2 packers do not
#             take (element count not a multiple of 4) runs on.
# (The round-1 "bf16x3" mode -- three bf16 limbs, six products -- was retired in round 4: slower than f16x2, never re-tuned.)
CONV_MODE = os.environ.get("ONDA_CONV_MODE", "f16x2")

# "f16x2": both operands are split into their two limbs BEFOREHAND, as limb rows in HBM (by the producing kernel or one split
# pass); the conv kernels move them to LDS by LDS-DMA only (csrc/conv_l2.hip).  (The round-1 kernels that split the
# activations inside the conv kernel -- ONDA_H2_PATH=reg -- were removed in round 5; the name stays for tools and tests.)
H2_PATH = "dma"

# the multi-GPU gradient exchange installs a callable here: called with the weight Parameter as soon as its gradient of the
# current backward pass has been accumulated in place (autograd's post-accumulate hook fires for it as well, later: the
# exchange counts a parameter once)
GRAD_READY = None

# bench.py sets this to a list to collect (kernel family, algorithmic flops, start event, end event)
# around every conv launch; the events are recorded on the launch stream (torch's current stream)
PROFILE = None



# ------------------------------------------------------------------------------- conv plumbing
# A device int32 the convolutions launched inside ``predicated(flag)`` carry as their predicate: each launch does nothing
# when the flag is 0 (the dynamic model's forward pass under the device-side hybrid switch).  Only the pre-split kernels
# honour it; the other conv modes refuse a predicate (EINVAL) rather than ignore it.
PREDICATE = None

# Row groups: inside ``with ops.row_groups(n):`` the first n images of every batch that passes a train-mode BatchNorm are
# one micro-batch and the rest another -- each normalised with its OWN batch statistics, only the second one moving the
# running statistics (the student's source-replay pass under BN_POLICY "freeze" and its target pass, prototypes.py:418-450,
# as ONE pass over both batches: every convolution, every weight gradient and every BatchNorm reduction is launched once
# instead of twice).  Only the pre-split kernels know about groups; the other paths raise.
ROW_GROUPS = 0

LIMB_ONLY = True  # (tests flip it to compare with fp32 outputs + split passes)

SHARE_GRADS = True  # tests turn this off to compare with autograd's own accumulation

# BatchNorm's small statistics passes inside the big launches that need them (csrc/norm_l2.hip: bn_finalize_l2 in the apply
# launch, bn_bwd_sums_l2 in the backward apply launch): 156 launches per adaptation step fewer -- and 27 ms per step SLOWER
# (A/B on one box, twice: 94.1 / 94.1 ms with the separate launches, 121.5 / 121.7 ms fused; gpurun_out/r06_a_*):
# the ~2 000 resident workgroups that wait for the first 64 poll ONE line at device scope, across the 8 XCDs' fabric, and the
# finalizing workgroups' own table reads and atomics queue behind the polls (175 us per fused launch against 9 us for the
# launch it removes).  Built, correct (the whole GPU suite passes with it on), OFF by default; ONDA_FUSE_BN=1 /
# tools/ab_flag.py onda_amd.ops FUSE_BN_FINALIZE True -- ... to measure it again.
FUSE_BN_FINALIZE = os.environ.get("ONDA_FUSE_BN", "0") == "1"
