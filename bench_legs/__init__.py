"""The legs of bench.py, one module per workload (What's weak 6: ten legs in one file).  bench.py keeps the contract: argument parsing, the headline
timing, the JSON line; every other workload it reports lives here and is imported by name."""
