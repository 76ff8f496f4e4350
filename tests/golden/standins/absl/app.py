"""TEST-ONLY stand-in for absl.app (see absl/__init__.py)."""
import sys


def run(main, argv=None):
  main(argv if argv is not None else sys.argv)
