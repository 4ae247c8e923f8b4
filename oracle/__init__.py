"""CPU oracle for the co-training hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch fp32 restatement (PyTorch-CPU, ATen kernels) of the
reference's per-step co-training path.  It exists to *check* the HIP product path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
    leg may import it;
  * the product package never imports it and has no CPU fallback -- the HIP path
    fails loudly when its extension is missing.

Where the arithmetic lives: the reference delegates every op to third-party PyTorch
(un-pinned; ``/root/reference/setup.py:6-14``).  The oracle therefore uses the same
ATen CPU kernels (torch 2.10.0 here) called from its own code, and is *pinned* against
golden vectors captured from the unmodified reference imported in the build container
(``tools/capture_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).  Each function cites the reference file:line it restates.
"""
from .nets import UNet, Enet, build_net, init_weights  # noqa: F401
from .losses import (cross_entropy_2d, entropy_2d, jsd_2d, kl_divergence_2d,  # noqa: F401
                     softmax_channels)
from .fgsm import fgsm_generate  # noqa: F401
from .adam import adam_reference_step  # noqa: F401
from .schedule import ramp_value  # noqa: F401
from .dice import dice_2d, dice_3d  # noqa: F401
from .step import OracleModel, cotrain_step  # noqa: F401
from .ensemble import soft_vote, hard_vote  # noqa: F401
