"""MI355X-native differentiable Gaussian rasterizer with a semantic-feature channel."""
