"""CPU oracle for the Foreground_Instance_Colorization hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sketchyscenecolorization_amd/`` may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / baseline.

PARITY UNPINNED at the TensorFlow boundary: the reference ships no tests,
golden vectors or checkpoints and TensorFlow is not installable here, so this
restatement follows the reference source line by line plus the published TF1
op definitions.  The only piece pinned against the real reference is
``text_processing`` (goldens under ``tests/golden/``).
"""
