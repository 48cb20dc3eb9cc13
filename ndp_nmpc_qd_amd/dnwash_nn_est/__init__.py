from .downwash_nn import DownwashNN
