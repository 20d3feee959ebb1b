"""TEST INFRASTRUCTURE (oracle harness): apriltags_eth is imported by
multical/board/aprilgrid_detector.py:1; only the pure-python corner geometry
(`AprilGridDetector.get_tag_corners_for_id`, :44-55) is used by the oracle, never the detector."""


def make_default_detector():
  return None
