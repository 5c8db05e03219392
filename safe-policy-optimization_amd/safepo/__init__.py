"""safepo (MI355X-native hot path).

Drop-in for the reference's single-agent collect -> GAE -> PPO-Lagrangian / CPO update path
(same module names: safepo.single_agent.ppo_lag, safepo.common.buffer, ...), implemented as
Python host code over a C ABI (include/safepo_hip.h) into hand-written HIP kernels for gfx950.
There is NO CPU fallback: every compute entry point raises if libsafepo_hip.so is missing.
"""
__version__ = "0.1.0"
