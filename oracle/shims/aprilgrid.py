"""TEST INFRASTRUCTURE (oracle harness): stand-in so multical/board/aprilgrid.py:16-22 `import aprilgrid`
succeeds quietly (tag-family tables are detection-only and out of scope)."""
