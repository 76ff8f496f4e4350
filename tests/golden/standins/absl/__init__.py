"""TEST-ONLY stand-in for the absent `absl` package (configuration plumbing only).

Used solely by tests/golden/make_fixtures.py, in the build container, so that the
reference's env modules (which do `from absl import app, flags`) can be imported
to generate golden vectors.  It holds flag names/defaults; it computes nothing
that ends up in a fixture.  Never imported by the product or on the GPU box.
"""
from . import flags  # noqa: F401
from . import app    # noqa: F401
