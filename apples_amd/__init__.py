"""MI355X-native implementation of the APPLES per-query hot path (distance vector +
least-squares placement sweep) behind run_apples.py's CLI and jplace output."""
__version__ = '0.1.0'
