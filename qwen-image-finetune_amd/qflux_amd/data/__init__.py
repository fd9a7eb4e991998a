from .cache_loader import (CachedEmbeddingDataset, collate_cached, convert_img_shapes_to_latent_space, pad_to_max_shape,  # noqa: F401
                           PrefetchLoader, write_cache_sample)
