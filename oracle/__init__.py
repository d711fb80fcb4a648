"""oracle/ -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT.

CPU restatement (PyTorch-CPU, fp32 or fp64, op-for-op) of the reference's
depth/pose training hot path (goodgodgd/xpt-mde-2021): view synthesis,
photometric L1 / SSIM, edge-aware smoothness, the pose algebra and the
multi-scale helpers.  Every function cites the reference file:line it follows
(paths relative to the reference checkout).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / reported baseline only.
The product package (``xpt_mde_2021_amd``) never imports it; its ops raise when
the HIP extension is missing.

Parity status
-------------
* The reference is TensorFlow 2.4 code.  TensorFlow is not installed in the
  build container (ordinary ``ModuleNotFoundError``), so the reference cannot
  be executed here and no reference-generated vectors exist.
* PINNED by the reference's own data-free known-answer tests (restated in
  ``tests/test_oracle_known_answers.py``): ``scale_intrinsic``, ``pixel2cam``,
  ``transform_to_source``, the bilinear neighbour weights / validity mask /
  reconstruction (``model/synthesize/test_synthesizing.py:149-301``), the twist
  <-> matrix conversions (``utils/convert_pose.py:222-271``,
  ``utils/tests.py:62-76``) and the 3x3 SAME average pool
  (``model/loss_and_metric/losses.py:541-559``).
* PARITY UNPINNED (no golden values or gradients exist anywhere in the
  reference): the numeric values of the L1 / SSIM / smoothness losses and all
  gradients.  They are checked by fp64 gradcheck of this restatement and by
  hand-computed cases only.  The NASNet-Mobile encoder arithmetic lives in
  ``tensorflow==2.4.1`` (``tf.keras.applications.nasnet``), absent from the
  reference checkout: parity unpinned (structural pins only).

The same op decomposition as the TF graph is kept on purpose (4 gathers +
stacked [B,N,4,HW,3] tensor in the sampler, 5 SAME average pools + tiled target
in SSIM, one call per scale) so that this package doubles as the
"reference-equivalent CPU path" timed by ``bench.py``'s ``cpu_baseline``.
"""
