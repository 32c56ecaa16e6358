"""MI355X-native APD-GICP scan matching for RIV-SLAM (hot path only)."""
