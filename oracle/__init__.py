"""CPU oracle of the MoE hot path — TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm (routers, dispatch index, expert FFN, combine, cache policy,
offload-directory format) used as the checker by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline / parity legs.  Nothing in the product package (moe-infinity_amd/) imports it, and the product has
no CPU path: without the HIP library and a GPU it raises.
"""
