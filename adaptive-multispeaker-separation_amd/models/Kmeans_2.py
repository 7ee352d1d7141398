# -*- coding: utf-8 -*-
"""Batched k-means mask assignment (reference models/Kmeans_2.py), host mirror over the HIP kernels."""
from ams_hip.kmeans_host import KMeans  # noqa: F401
