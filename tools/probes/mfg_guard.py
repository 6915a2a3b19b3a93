"""Kept under its round-3 name: the bounds-checked replay now covers manufacturing, hospital and fleet — tools/probes/guard_run.py."""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard_run.py"), run_name="__main__")
