"""MI355X-native particle-filter point-cloud tracker (drop-in for the PCL 1.8.0 path that
cmaestre/pcl_tracking drives from src/auto_tracking.cpp)."""
__version__ = "0.1.0"
