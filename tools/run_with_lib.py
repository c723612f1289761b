"""Runs bench.py against another build of the library (measurement builds under tools/bin/):
python tools/run_with_lib.py <path to libdpf*.so> [bench.py arguments]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import similaritysearchbyrdf_amd._lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench

bench.main()
