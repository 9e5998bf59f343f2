import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import tests.test_gpu_split as t
bad = 0
for seed in range(6, 66):
    try:
        t.test_random_shapes_split_vs_fp32(seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:200])
print("done, failures:", bad)
