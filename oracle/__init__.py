"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity status: parity unpinned (no TensorFlow, no reference golden vectors).
"""
