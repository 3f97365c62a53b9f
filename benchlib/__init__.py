"""Parts of bench.py (the repo-root CLI keeps its flags and its one JSON line; see its docstring):

    workload      BASELINE.json configs, which view a rank renders at a step, the algorithmic-bytes model, kernels_sha
    timing        interval arithmetic behind gpu_ms_per_step_timed, warm-up constants
    roofline      the `roofline` object: HBM fraction of the dominant kernel, counter traffic and SIMD occupancy from profiles/pmc_traffic.json
    cpu_baseline  the `cpu_baseline` object: the plain-C oracle timed on the host cores this process may use
    dist          self-launch under torch.distributed.run, the child processes for side figures (drop-in probes, world-1 gather anchor)
"""
