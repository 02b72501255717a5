"""cProfile of bg_colorization_main.py --mode train (synthetic scenes, 768 x 768): where the loop's own thread spends an iteration"""
import cProfile
import os
import pstats
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(tempfile.mkdtemp())
import bg_colorization_main as bg       # noqa: E402
pr = cProfile.Profile()
pr.enable()
bg.main(['--mode', 'train', '--image_size', '768', '--max_steps', '200', '--save_freq', '100000', '--progress_freq', '100',
         '--summary_freq', '100000'])
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
