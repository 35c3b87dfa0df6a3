"""CPU oracle for the MolGym PPO policy/value hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``molgym_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker.

What it is: a plain-PyTorch (CPU, float32 or float64) restatement of
``ppo.train -> ppo.compute_loss -> CovariantAC.step(obs, actions) -> backward``
at the operator granularity of the reference (per-(l1, l2) Python loops inside
the Clebsch-Gordan products, the Lebedev grid re-evaluated on every call, the
float64 loss seam of ``molgym/ppo.py:28-33``).

Pinning status (see DESIGN.md "Oracle"):
  * in-tree reference code that imports in the build container (``ppo``,
    ``buffer``, ``modules``, ``gmm``, ``so3_tools``, ``spherical_dists``) is
    PINNED: ``oracle/make_golden.py`` imports it from /root/reference and the
    vectors are committed under ``tests/golden``.
  * the third-party encoder arithmetic (cormorant @ 6a4b6370, torch-scatter
    2.0.5, quadpy 0.16.2) is NOT on disk anywhere.  Its layers are restated
    from the published algorithm; their numerical outputs are
    **PARITY UNPINNED** except for the reference's own known answers
    (Y_lm values of tests/agents/covariant/test_sphs.py, masked_softmax sums
    of tests/test_modules.py) and symmetry properties (rotation invariance).
"""
