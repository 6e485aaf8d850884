from .monte_carlo import monte_carlo

__all__ = ["monte_carlo"]
