"""pta_replicator_amd - MI355X-native stochastic-signal injection for pulsar-timing-array simulations.

Drop-in for the hot path of bencebecsy/pta_replicator (add_measurement_noise / add_jitter / add_red_noise /
add_gwb / add_cgw) on AMD Instinct MI355X: Python keeps the pulsar bookkeeping, hand-written HIP kernels behind a
ctypes C ABI (include/pta_replicator_amd.h) do the arithmetic.  ``ReplicaEngine`` batches independent
realisations on one GPU and shards them across GPUs.  There is no CPU fallback.
"""
__version__ = "0.1.0"
