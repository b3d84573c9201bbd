"""srbd_horizon_amd -- MI355X-native DDP solver for the SRBD centroidal-MPC hot path of hucebot/srbd_horizon.

Only what the hot path needs: the C-ABI HIP engine (``csrc/``, ``include/sddp.h``), its ctypes
binding (``_lib``), the batched engine object (``engine``), and host-side mirrors of the reference surfaces on either
side of the solve: ``ddp.DDPSolver`` (python/ddp.py), ``prb`` builders (python/prb.py), ``wpg.steps_phase``
(python/wpg.py), the receding-horizon loop (``mpc``, python/dsrbd_example.py) and instance sharding (``dist``).
"""
__version__ = "0.1.0"
