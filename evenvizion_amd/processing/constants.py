"""Tunables of the hot path; same names and values as the reference (evenvizion/processing/constants.py:15-32)."""

#: coordinates at or above this value are treated as undefined (constants.py:15)
INFINITY_COORDINATE = 10000

#: minimum inlier share of the final RANSAC for H to be accepted (constants.py:19, used at utils.py:359)
LENGTH_ACCOUNTED_POINTS = 0.7

#: reprojection threshold handed to findHomography (constants.py:22)
THRESHOLD_FOR_FIND_HOMOGRAPHY = 3.0

#: Lowe's ratio (constants.py:25)
LOWES_RATIO = 0.5

#: minimum number of matches after the ratio / one-to-one filters (constants.py:28)
MINIMUM_MATCHING_POINTS = 4

#: heat-map normalisation constant (constants.py:32; visualisation only, kept for API completeness)
HEATMAP_CONSTANT = 1000
