"""CPU oracle for the occm hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker.  The product path (``occm_amd``)
calls hand-written HIP kernels through ``libocc_hip.so`` and raises when that
library is missing; it never falls back to this package.

Every function cites the reference file:line (relative to the upstream
``nguyenvulong/occm`` checkout) whose arithmetic it restates.

Pinning status (see DESIGN.md "Oracle"):
  * rawboost_np, losses_ref, eer_ref, aasist_ref, senet_ref: pinned against the
    reference's own Python modules run in the build container; vectors in
    ``tests/golden/`` (generator: ``oracle/gen_golden.py``).
  * xlsr_ref: the arithmetic lives in third-party fairseq @ a540213 which is
    not vendored in the reference and not installed -> **parity unpinned** vs
    the reference.  It is pinned only against HuggingFace ``Wav2Vec2Model``
    (an independent restatement of the same architecture).
"""
