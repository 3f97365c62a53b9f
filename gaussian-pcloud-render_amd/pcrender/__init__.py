"""Host-side mirror of the reference caller for the rasterizer hot path (camera/settings construction,
synthetic workloads, view sharding).  Python, as the reference's caller is Python."""
