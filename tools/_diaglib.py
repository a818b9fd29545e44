"""Diagnostic tools only: run against another build of the library.  The product (dragposer_amd/_lib.py) reads no environment variable;
the tools under tools/ honour DRAGPOSER_LIB by pointing the binding at that file before the first context is created."""
import os


def use_env_library():
    path = os.environ.get("DRAGPOSER_LIB")
    if path:
        from dragposer_amd import _lib

        _lib.LIB_PATH = os.path.abspath(path)
    return path
