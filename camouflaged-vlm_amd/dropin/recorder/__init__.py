"""Drop-in for the reference's `recorder` package, evaluation classes only (recorder/__init__.py:4 exports
OVCOSMetricer; test_ovcos_maskdecoder_edge.py:18 imports recorder.new_evaluator.Classification).  Training-time
helpers (TrainingCounter, HistoryBuffer, plot_results) are outside the path."""
from .ovcos_metricer import OVCOSMetricer  # noqa: F401
