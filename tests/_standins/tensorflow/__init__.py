"""Inert stand-in: the reference imports tensorflow in util.py/generate.py but
only dereferences it inside model.py / visualize.py.  Build-owned
test helper, used only by tests/golden/make_golden.py in the authoring container."""
