"""maskrcnn_benchmark.data: only the input TRANSFORMS are on the MI355X path; datasets, samplers and
collators stay the reference's (out of scope)."""
