"""legs of bench.py that are long enough to live on their own (round 6: the offload regime; the drop-in leg is tools/dropin_time.py)"""
