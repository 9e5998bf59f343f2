"""Parts of bench.py (the repo-root contract script): the CPU baselines, BASELINE configs 4 and 5, shared helpers."""
