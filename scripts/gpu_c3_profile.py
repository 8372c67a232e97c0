"""Dev script: cProfile of the CLI on one large image (see gpu_c3_large_image.py)."""
import cProfile, pstats, sys, io
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path(str(__import__('pathlib').Path(__file__).resolve().parent / 'gpu_c3_large_image.py'), run_name='__main__')
finally:
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
