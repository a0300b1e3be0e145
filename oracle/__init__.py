"""CPU oracle for the burst super-resolution hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker.  The product
(``handheld-multi-frame-super-resolution_amd/``) never imports it and has no
CPU fallback.

What it is: a NumPy restatement of the reference's algorithm for the path
``main()`` of ``handheld_super_resolution/super_resolution.py:41-200`` and
everything it calls, following the reference's *Numba typing*
(which operations run in float64 and where values are rounded to float32 —
SURVEY.md App. B) and its deterministic quirks (App. A).  Every function cites
the reference file:line it restates.

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md §4), so this oracle is pinned against outputs of the reference's
own code executed in the build container (``tools/refsim``: the reference's
kernel bodies run under a CPU stand-in for ``numba.cuda``).  The captured
vectors live in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every stage and two end-to-end bursts (x2 RGGB; x1 BGGR with white balance and
the accumulated-robustness denoiser) against them — measured end-to-end
difference to the reference's own output 5e-6.  The level-0 ``L1`` search is
undefined behaviour upstream (App. A D1) and is therefore *unpinned*: the
oracle defines its intended semantics.  ``frontend`` (burst normalisation,
Monte-Carlo noise curves — SURVEY.md §8f-3) restates code whose reference
counterpart needs rawpy / draws unseeded random numbers: the normalisation is
pinned by hand-computed known answers, the Monte-Carlo only statistically.
"""
from .params import update_snr_config, sanitize_config, lerp  # noqa: F401
from .grey import compute_grey_images, grey_fft, decimate_to_grey  # noqa: F401
from .pyramid import gaussian_taps, downsample, build_gaussian_pyramid  # noqa: F401
from .align import (init_alignment, align, align_lvl, init_ica, bm_l2, bm_l1, ica, upscale_lvl,  # noqa: F401
                    level_shapes)
from .kernels import estimate_kernels, gat  # noqa: F401
from .robustness import (init_robustness, compute_robustness, guide_image, local_stats,  # noqa: F401
                         upscale_warp_stats, compute_s, local_min)
from .merge import merge, merge_ref, divide  # noqa: F401
from .pipeline import main  # noqa: F401
from .parallel import main_parallel, throughput_all_cores, available_cores  # noqa: F401
from . import frontend  # noqa: F401
from . import post  # noqa: F401
