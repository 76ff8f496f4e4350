"""TEST-ONLY stand-in for the absent `gymnasium` package: only utils.seeding.np_random.

See tests/golden/standins/absl/__init__.py for why this exists.  `import gymnasium`
style consumers (the reference's Gym/Zoo wrappers) are NOT served by this stub.
"""
