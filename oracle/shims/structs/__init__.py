"""TEST INFRASTRUCTURE (oracle harness): stand-in for the py-structs package. See struct.py."""
