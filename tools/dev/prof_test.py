import cProfile, pstats, sys
sys.path.insert(0, '.')
import tests.test_gpu_train as T
T.test_gradients_at_the_configs_batch("carpet", (1, 6), None, 0.0, 64, 64)   # warm
pr = cProfile.Profile(); pr.enable()
T.test_gradients_at_the_configs_batch("carpet", (1, 6), None, 0.0, 1024, 256)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
