from .crossdomain_recommender import CrossDomainRecommender  # noqa: F401
