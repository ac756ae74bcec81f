"""One-line activation of the drop-in boundary: with ``<repo>/dropin`` on PYTHONPATH the interpreter imports this
module at start-up (``site``), before the reference driver runs, and ``instascene_amd.dropin.install()`` makes
``diff_surfel_rasterization``, ``simple_knn._C``, ``gaussian_renderer.render`` and
``utils.contrastive_utils.contrastive_loss`` resolve to the HIP library (see instascene_amd/dropin.py).

    PYTHONPATH=/path/to/repo/dropin python train_semantic.py -s ...

Set ``ISR_DROPIN=0`` to leave the interpreter untouched.  Nothing heavy is imported here (no torch, no HIP): the
rebinding happens when the reference's modules are imported.
"""
import os
import sys

if os.environ.get("ISR_DROPIN", "1") != "0":
    _repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if _repo not in sys.path:
        sys.path.append(_repo)
    try:
        from instascene_amd.dropin import install as _install
        _install()
    except Exception as _e:      # never break interpreter start-up; the reference's imports then fail loudly on their own
        print(f"[instascene_amd] drop-in not installed: {_e!r}", file=sys.stderr)
