// host_types.hpp — the reference's host-side PODs without Godot.
//
// Same names, field order, sizes and constructor semantics as
// src/core/ray.h:25-98, src/core/intersection.h:16-61, src/core/triangle.h:22-51
// and src/core/stats.h:20-55 at precision=single, with a 3-float Vector3 in
// place of godot::Vector3.  These are the types RayDispatcher / GPURayCaster
// take in the reference; a maintainer swapping the backend keeps their own.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace mrt {

struct Vector3 {
	float x = 0.0f, y = 0.0f, z = 0.0f;
	Vector3() = default;
	Vector3(float p_x, float p_y, float p_z) : x(p_x), y(p_y), z(p_z) {}
	Vector3 operator+(const Vector3 &o) const { return Vector3(x + o.x, y + o.y, z + o.z); }
	Vector3 operator-(const Vector3 &o) const { return Vector3(x - o.x, y - o.y, z - o.z); }
	Vector3 operator*(float s) const { return Vector3(x * s, y * s, z * s); }
	float dot(const Vector3 &o) const { return x * o.x + y * o.y + z * o.z; }
	Vector3 cross(const Vector3 &o) const { return Vector3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
	float length_squared() const { return x * x + y * y + z * z; }
	Vector3 normalized() const
	{
		float l2 = length_squared();
		if (l2 == 0.0f) return Vector3();
		float l = std::sqrt(l2);
		return Vector3(x / l, y / l, z / l);
	}
	bool is_finite() const { return std::isfinite(x) && std::isfinite(y) && std::isfinite(z); }
};
static_assert(sizeof(Vector3) == 12, "Vector3 must be 3 floats");

// src/core/ray.h:25-98
struct Ray {
	Vector3 origin;
	Vector3 direction;
	Vector3 inv_direction;
	int dir_sign[3];
	float t_min;
	float t_max;
	uint32_t flags;

	Ray() : t_min(0.001f), t_max(FLT_MAX), flags(0) { dir_sign[0] = dir_sign[1] = dir_sign[2] = 0; }
	Ray(const Vector3 &o, const Vector3 &d, float t0 = 0.001f, float t1 = FLT_MAX)
		: origin(o), direction(d), t_min(t0), t_max(t1), flags(0) { _precompute(); }
	bool is_valid() const { return direction.is_finite() && origin.is_finite() && t_min <= t_max; }
	Vector3 at(float t) const { return origin + direction * t; }

private:
	// contract of src/core/ray.h:78-89: 1 / d per axis, a component with |d| < 1e-9 gets +-1e9 by the sign test d < 0
	// (so -0.0f gets +1e9, and a NaN stays a NaN); dir_sign = (d < 0)
	static float reciprocal_clamped(float c)
	{
		constexpr float tiny = 1e-9f;
		if (!(std::fabs(c) < tiny)) return 1.0f / c;
		const float huge = 1.0f / tiny;
		return c < 0.0f ? -huge : huge;
	}
	void _precompute()
	{
		const float c[3] = { direction.x, direction.y, direction.z };
		float r[3];
		for (int a = 0; a < 3; a++) { r[a] = reciprocal_clamped(c[a]); dir_sign[a] = c[a] < 0.0f ? 1 : 0; }
		inv_direction = Vector3(r[0], r[1], r[2]);
	}
};
static_assert(sizeof(Ray) == 60, "Ray must be 60 bytes");

// src/core/intersection.h:16-61
struct Intersection {
	float t;
	Vector3 position;
	Vector3 normal;
	float u;
	float v;
	uint32_t prim_id;
	uint32_t hit_layers;
	static constexpr uint32_t NO_HIT = UINT32_MAX;
	Intersection() : t(FLT_MAX), position(), normal(), u(0.0f), v(0.0f), prim_id(NO_HIT), hit_layers(0) {}
	void set_miss() // everything but position / normal back to the default record (intersection.h:45-52 leaves those two alone)
	{
		const Vector3 keep_p = position, keep_n = normal;
		*this = Intersection();
		position = keep_p; normal = keep_n;
	}
	bool hit() const { return prim_id != NO_HIT; }
};
static_assert(sizeof(Intersection) == 44, "Intersection must be 44 bytes");

// src/core/triangle.h:22-51 (constructor only; intersection runs on the device)
struct Triangle {
	Vector3 v0, v1, v2;
	Vector3 edge1, edge2, normal;
	uint32_t id;
	uint32_t layers;
	Triangle() : id(0), layers(0xFFFFFFFF) {}
	Triangle(const Vector3 &a, const Vector3 &b, const Vector3 &c, uint32_t p_id, uint32_t p_layers = 0xFFFFFFFF)
		: v0(a), v1(b), v2(c), edge1(b - a), edge2(c - a), normal(edge1.cross(edge2).normalized()), id(p_id), layers(p_layers) {}
};
static_assert(sizeof(Triangle) == 80, "Triangle must be 80 bytes");

// src/core/stats.h:20-55
struct RayStats {
	uint64_t rays_cast = 0;
	uint64_t tri_tests = 0;
	uint64_t bvh_nodes_visited = 0;
	uint64_t hits = 0;
	void reset() { rays_cast = tri_tests = bvh_nodes_visited = hits = 0; }
	RayStats &operator+=(const RayStats &o)
	{
		rays_cast += o.rays_cast; tri_tests += o.tri_tests; bvh_nodes_visited += o.bvh_nodes_visited; hits += o.hits;
		return *this;
	}
};

} // namespace mrt
