"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float64 / plain C) of the reference hot path.

Nothing in the product package (``srbd_horizon_amd``) may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED: the reference's DDP arithmetic lives in the external native module ``pyddp`` fed by
CasADi/Horizon graphs (reference ``python/ddp.py:1``, ``:93-94``, ``:101``); none of them is vendored,
pinned or installed, and the reference has no tests or golden vectors.  This oracle therefore restates the
*problem definition* that is in the tree (``python/prb.py``, ``python/ddp.py:165-230``, ``python/wpg.py``)
and a documented textbook multiple-shooting DDP ("Algorithm").  It is pinned by
(1) contact-schedule fixtures generated from the reference's own ``wpg.py`` (``tests/golden/wpg_*.npz``),
(2) sympy symbolic differentiation and complex-step finite differences of the dynamics/residuals,
(3) the LIP LQ known answer (dense KKT solve), (4) optimality residuals.
"""
