"""TEST INFRASTRUCTURE (oracle harness): numpy-quaternion is only *imported* on the BA path
(/root/reference/multical/transform/interpolate.py:2); nothing on the path calls it."""
