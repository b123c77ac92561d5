"""loner_amd -- MI355X-native mapping hot path for LONER-style LiDAR neural SLAM."""
__version__ = "0.1.0"
