"""Puts the product package directory on sys.path.

The package directory is named after the reference repository
(`temporally-consistent-stereo-matching_amd/`), which is not a valid Python identifier, so the
importable packages live inside it: `core` (the drop-in mirror of the reference's `core` package)
and `tcs_mi355` (native binding, harness, synthetic data, weights).
"""
import os
import sys

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(REPO_ROOT, "temporally-consistent-stereo-matching_amd")


def add_product_path():
    if PKG_DIR not in sys.path:
        sys.path.insert(0, PKG_DIR)
    return PKG_DIR
