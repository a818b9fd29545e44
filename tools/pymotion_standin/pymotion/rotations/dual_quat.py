"""placeholder (off the hot path; never called by tools/make_goldens.py)"""
