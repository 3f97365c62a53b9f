"""Timing helpers of bench.py."""

GC_RECOVER_S = 0.15   # untimed load between the garbage collection and the first timed block (seconds)


def union_ms(intervals):
    """total length of the union of (start, end) intervals"""
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in sorted(intervals):
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot
