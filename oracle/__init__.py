"""CPU oracle for the UFM dense-correspondence inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ufm_amd/`` (the product) may import,
call, link or execute anything from this package.  The only legitimate users
are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker, never as the thing measured or
shipped.

Parity status (see DESIGN.md "Oracle"):
  * glue (pre/post-processing, wiring, classification refinement): PINNED by
    golden vectors captured from the reference's own code run in the build
    container with the absent third-party imports stubbed
    (``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``).
  * third-party ``uniception`` blocks (DINOv2 encoder, global-attention
    info-sharing, DPT heads, adaptors): the source is NOT in ``/root/reference``
    (empty git submodule, version unpinned).  They are restated here from the
    published architectures; the encoder is cross-checked against
    ``transformers.Dinov2Model`` built offline from a config object.  For these
    blocks: **parity unpinned**.
"""
