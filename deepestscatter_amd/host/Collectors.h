// Collectors.h -- host-side mirror of the reference's dataset collectors on top of the libcloudtrace C ABI:
//   ScatterSampleCollector     src/Scene/ScatterSampleCollector.{h,cpp}   -> ct_generate_scatter_samples
//   RadianceCollector          src/Scene/RadianceCollector.{h,cpp}        -> ct_point_radiance_launch
//   DisneyDescriptorCollector  src/Scene/DisneyDescriptorCollector.{h,cpp} -> ct_collect_descriptors
//   Dataset                    src/Util/Dataset/Dataset.{h,cpp}            (named tables, int32 keys, proto3 values)
// Same class names, the same init / update / isCompleted life cycle (SceneItem), the same batch numbering
// (BatchSettings: batchStartId = sceneId * 2048, batchSize 2048, Tasks.cpp:136-137) and the same host arithmetic
// (task replication, PointRadianceTask::operator+=, the 95 % confidence-interval convergence rule).  The device work
// happens behind the C ABI.  The reference stores its records in LMDB; neither liblmdb nor protobuf's C++ runtime is on
// the target image, so `Dataset` here writes one flat file per table -- magic, table name, record count, then (int32
// key, uint32 length, proto3 bytes) per record -- with hand-written encoders for the four small messages
// (DeepestScatter_Train/Protocols/*.proto); tools/flat_to_lmdb.py turns those files into the reference's LMDB layout
// where the `lmdb` module exists.  The Python mirror (deepestscatter_amd/collector.py) writes the same bytes; the tests
// compare the two.
#pragma once

#include <chrono>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <map>

#include "Scene.h"

namespace DeepestScatter
{
    struct BatchSettings                                                        // src/Util/Dataset/BatchSettings.h
    {
        BatchSettings(uint32_t batchStartId, uint32_t batchSize) : batchStartId(batchStartId), batchSize(batchSize) {}
        uint32_t batchStartId, batchSize;
    };

    // ---- proto3 wire format of the messages this path exchanges (DeepestScatter_Train/Protocols/*.proto) ----------
    namespace Persistance
    {
        inline void putVarint(std::string& out, uint64_t v)
        {
            while (v >= 0x80) { out.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
            out.push_back((char)v);
        }
        inline void putFloatField(std::string& out, int field, float v)       // proto3 omits default (zero) values
        {
            if (v != 0.0f) { out.push_back((char)(field << 3 | 5)); out.append(reinterpret_cast<const char*>(&v), 4); }
        }
        inline std::string vector3(const float v[3])                          // Vector3 { float x = 1, y = 2, z = 3 }
        {
            std::string b;
            for (int k = 0; k < 3; k++) putFloatField(b, k + 1, v[k]);
            return b;
        }
        inline void putMessageField(std::string& out, int field, const std::string& body)
        {
            out.push_back((char)(field << 3 | 2));
            putVarint(out, body.size());
            out += body;
        }
        // Result { float light_intensity = 1; bool is_converged = 2; }            Result.proto:5-8
        inline std::string result(float lightIntensity, bool isConverged)
        {
            std::string b;
            putFloatField(b, 1, lightIntensity);
            if (isConverged) { b.push_back('\x10'); b.push_back('\x01'); }
            return b;
        }
        // ScatterSample { int32 scene_setup_id = 1; Vector3 point = 2; Vector3 view_direction = 3; }
        inline std::string scatterSample(int32_t sceneSetupId, const float point[3], const float viewDirection[3])
        {
            std::string b;
            if (sceneSetupId != 0) { b.push_back('\x08'); putVarint(b, (uint64_t)(int64_t)sceneSetupId); }
            putMessageField(b, 2, vector3(point));
            putMessageField(b, 3, vector3(viewDirection));
            return b;
        }
        // SceneSetup { string cloud_path = 1; float cloud_size_m = 2; Vector3 light_direction = 3; }
        inline std::string sceneSetup(const std::string& cloudPath, float cloudSizeM, const float lightDirection[3])
        {
            std::string b;
            if (!cloudPath.empty()) putMessageField(b, 1, cloudPath);
            putFloatField(b, 2, cloudSizeM);
            putMessageField(b, 3, vector3(lightDirection));
            return b;
        }
        // DisneyDescriptor { bytes grid = 1; }                                     DisneyDescriptor.proto:7-10
        inline std::string disneyDescriptor(const uint8_t* grid, size_t bytes)
        {
            std::string b;
            putMessageField(b, 1, std::string(reinterpret_cast<const char*>(grid), bytes));
            return b;
        }
        // the one decoder the pipeline needs: RadianceCollector and DisneyDescriptorCollector read ScatterSample records
        inline void readScatterSample(const std::string& b, float point[3], float viewDirection[3])
        {
            for (int k = 0; k < 3; k++) point[k] = viewDirection[k] = 0.f;
            size_t i = 0;
            auto varint = [&]() { uint64_t v = 0; int s = 0; while (true) { const uint8_t c = (uint8_t)b.at(i++); v |= (uint64_t)(c & 0x7f) << s; if (!(c & 0x80)) return v; s += 7; if (s > 63) throw std::runtime_error("varint longer than ten bytes in a ScatterSample record"); } };
            while (i < b.size())
            {
                const uint8_t tag = (uint8_t)b[i++];
                if (tag == 0x08) { varint(); continue; }
                if (tag != 0x12 && tag != 0x1a) throw std::runtime_error("unexpected tag in a ScatterSample record");
                float* dst = tag == 0x12 ? point : viewDirection;
                const size_t n = (size_t)varint();
                if (n > b.size() - i) throw std::runtime_error("Vector3 longer than its ScatterSample record");
                const size_t end = i + n;
                while (i < end)
                {
                    const int k = ((uint8_t)b.at(i) >> 3) - 1;
                    if (k < 0 || k > 2 || i + 5 > b.size()) throw std::runtime_error("malformed Vector3 in a ScatterSample record");
                    std::memcpy(&dst[k], b.data() + i + 1, 4);
                    i += 5;
                }
            }
        }
    }

    // The collectors log like the reference's (std::cout); with several scene setups in flight (`cloudtrace collect --jobs`)
    // the per-update lines of different scenes would interleave, so they can be switched off.
    struct CollectorLog
    {
        static bool& quiet() { static bool q = false; return q; }
        static std::ostream& out()
        {
            static std::ostream null(nullptr);                                    // no buffer: every write is dropped
            return quiet() ? null : std::cout;
        }
    };

    // Dataset (Dataset.h:20-110): one named table per message type, int32 record ids.  batchAppend / getRecord like the
    // reference; `save` writes the flat stand-in for the LMDB environment (see the header comment).
    class Dataset
    {
    public:
        void batchAppend(const std::string& table, const std::vector<std::string>& records, int32_t startId)   // Dataset.h:60-75
        {
            auto& t = tables[table];
            for (size_t i = 0; i < records.size(); i++) t[startId + (int32_t)i] = records[i];
        }
        const std::string& getRecord(const std::string& table, int32_t id) const                                  // Dataset.h:39-58
        {
            const auto t = tables.find(table);
            if (t == tables.end() || !t->second.count(id)) throw std::runtime_error("no record " + std::to_string(id) + " in table " + table);
            return t->second.at(id);
        }
        void merge(const Dataset& other)                                                                          // the tables of another scene setup (disjoint record ids)
        {
            for (const auto& [name, records] : other.tables)
                for (const auto& [id, bytes] : records) tables[name][id] = bytes;
        }
        size_t getRecordsCount(const std::string& table) const { const auto t = tables.find(table); return t == tables.end() ? 0 : t->second.size(); }

        void save(const std::filesystem::path& dir) const
        {
            std::filesystem::create_directories(dir);
            for (const auto& [name, records] : tables)
            {
                std::ofstream f(dir / (name + ".flat"), std::ios::binary);
                const uint32_t nameLen = (uint32_t)name.size(), count = (uint32_t)records.size();
                f.write("DSFLAT1\0", 8);
                f.write(reinterpret_cast<const char*>(&nameLen), 4);
                f.write(name.data(), nameLen);
                f.write(reinterpret_cast<const char*>(&count), 4);
                for (const auto& [id, bytes] : records)
                {
                    const uint32_t len = (uint32_t)bytes.size();
                    f.write(reinterpret_cast<const char*>(&id), 4);
                    f.write(reinterpret_cast<const char*>(&len), 4);
                    f.write(bytes.data(), len);
                }
            }
        }

    private:
        std::map<std::string, std::map<int32_t, std::string>> tables;
    };

    // The renderer handle for a collection task: the same CtScene the scene items publish, no frame to speak of
    // (EmptyRenderer in the reference, Tasks.cpp:139).
    inline void createCollectorHandle(Context& context, int estimator = CT_EST_MARCH)
    {
        CtScene& s = context.scene;
        s.density_host = context.density.data();
        s.mie_host = context.mie.data();
        s.chopped_mie_host = context.choppedMie.data();
        s.mie_count = (uint32_t)context.mie.size();
        s.estimator = estimator;
        if (s.width == 0 || s.height == 0) { s.width = 64; s.height = 64; }
        context.create();
    }

    class ScatterSampleCollector : public SceneItem                            // ScatterSampleCollector.cpp:10-78
    {
    public:
        ScatterSampleCollector(std::shared_ptr<Context> context, std::shared_ptr<Dataset> dataset, BatchSettings settings, int32_t sceneSetupId)
            : context(std::move(context)), dataset(std::move(dataset)), settings(settings), sceneSetupId(sceneSetupId) {}

        void init() override {}
        void update() override                                                  // reset() + collect(), :22-62
        {
            if (done) return;
            CollectorLog::out() << "Generating samples..." << std::endl;
            std::vector<float> positions(3 * (size_t)settings.batchSize), directions(3 * (size_t)settings.batchSize);
            // one launch over batchSize threads; the batch's first record id seeds it (the reference mixes in clock())
            Context::check(ct_generate_scatter_samples(context->handle, settings.batchSize, settings.batchStartId, positions.data(), directions.data()),
                           context->handle, "ct_generate_scatter_samples");
            CollectorLog::out() << "Serializing samples..." << std::endl;
            std::vector<std::string> samples(settings.batchSize);
            for (uint32_t i = 0; i < settings.batchSize; i++)
                samples[i] = Persistance::scatterSample(sceneSetupId, &positions[3 * (size_t)i], &directions[3 * (size_t)i]);
            CollectorLog::out() << "Writing samples..." << std::endl;
            dataset->batchAppend("ScatterSample", samples, (int32_t)settings.batchStartId);
            CollectorLog::out() << "Finished writing samples." << std::endl;
            done = true;
        }
        bool isCompleted() override { return done; }

    private:
        std::shared_ptr<Context> context;
        std::shared_ptr<Dataset> dataset;
        BatchSettings settings;
        int32_t sceneSetupId;
        bool done = false;
    };

    class RadianceCollector : public SceneItem                                 // RadianceCollector.cpp:15-192
    {
    public:
        static constexpr uint32_t MAX_THREAD_COUNT = 10 * 2048;               // :17

        RadianceCollector(std::shared_ptr<Context> context, std::shared_ptr<Dataset> dataset, BatchSettings settings)
            : context(std::move(context)), dataset(std::move(dataset)), settings(settings) {}

        void init() override                                                    // :19-58
        {
            std::vector<CtPointRadianceTask> tasks;
            for (uint32_t i = 0; i < settings.batchSize; i++)
            {
                CtPointRadianceTask t{};
                t.id = (int32_t)i;
                Persistance::readScatterSample(dataset->getRecord("ScatterSample", (int32_t)(settings.batchStartId + i)), t.position, t.direction);
                tasks.push_back(t);
            }
            scheduleTasks(tasks);
            frameId = 0;
        }

        int32_t getConvergedCount() const { return (int32_t)convergedTasks.size(); }
        int32_t getRemainingCount() const { return (int32_t)settings.batchSize - getConvergedCount(); }

        void update() override                                                  // :73-141
        {
            if (allPixelsConverged) return;
            // 100 launches of estimateEmission over the replicated tasks, frame ids frameId+1 .. frameId+100 (:84-96)
            const auto t1 = std::chrono::steady_clock::now();
            Context::check(ct_point_radiance_launch(context->handle, tasksBuffer.data(), threadsCount, frameId + 1, 100), context->handle,
                           "ct_point_radiance_launch");
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            frameId += 100;
            totalRenderMs += ms;
            totalExperiments += (uint64_t)threadsCount * 100u;
            updates++;
            CollectorLog::out() << "MS/Render: " << ms << " " << settings.batchSize - (uint32_t)getConvergedCount() << std::endl;   // :97
            std::vector<CtPointRadianceTask> todoTasks;
            const uint32_t remaining = (uint32_t)getRemainingCount();
            for (uint32_t i = 0; i < remaining; i++)
            {
                CtPointRadianceTask& representative = tasksBuffer[(size_t)i * taskRepeatCount];
                for (uint32_t j = 1; j < taskRepeatCount; j++) merge(representative, tasksBuffer[(size_t)i * taskRepeatCount + j]);
                bool isConverged = relativeConfidenceInterval(representative) < 2e-2f || absoluteConfidenceInterval(representative) < 1e-4f;
                if (representative.radiance < FLT_EPSILON) isConverged = representative.experimentCount > 100000;
                (isConverged ? convergedTasks : todoTasks).push_back(representative);
            }
            allPixelsConverged = getConvergedCount() == (int32_t)settings.batchSize;
            CollectorLog::out() << "converged: " << getConvergedCount() << " of " << settings.batchSize << std::endl;
            if (allPixelsConverged) recordToDataset();
            else scheduleTasks(todoTasks);
        }

        bool isCompleted() override { return allPixelsConverged; }

        // what the reference's "MS/Render" lines add up to (RadianceCollector.cpp:87-97), for `cloudtrace collect --timings`
        double totalRenderMs = 0;
        uint64_t totalExperiments = 0;
        uint32_t updates = 0;

        // PointRadianceTask.h:23-36 (95 % confidence), :56-68 (operator+=: the M2 values are added as they are)
        static float absoluteConfidenceInterval(const CtPointRadianceTask& t)
        {
            const float N = (float)t.experimentCount;
            const float sigma = sqrtf(t.runningVariance / N);
            return 1.96f * sigma / sqrtf(N);
        }
        static float relativeConfidenceInterval(const CtPointRadianceTask& t) { return absoluteConfidenceInterval(t) / (t.radiance + FLT_EPSILON); }
        static void merge(CtPointRadianceTask& into, const CtPointRadianceTask& other)
        {
            if (other.id != into.id) throw std::invalid_argument("Different point radiance tasks cannot be merged into one!");
            const float newWeight = (float)other.experimentCount * 1.0f / (float)(into.experimentCount + other.experimentCount);
            into.radiance += (other.radiance - into.radiance) * newWeight;
            into.runningVariance += other.runningVariance;
            into.experimentCount += other.experimentCount;
        }

    private:
        void recordToDataset()                                                   // :148-169
        {
            std::sort(convergedTasks.begin(), convergedTasks.end(), [](const CtPointRadianceTask& a, const CtPointRadianceTask& b) { return a.id < b.id; });
            CollectorLog::out() << "Serializing emissions..." << std::endl;
            std::vector<std::string> results(settings.batchSize);
            for (uint32_t i = 0; i < settings.batchSize; i++) results[i] = Persistance::result(convergedTasks[i].radiance, true);
            CollectorLog::out() << "Writing emissions..." << std::endl;
            dataset->batchAppend("Result", results, (int32_t)settings.batchStartId);
            CollectorLog::out() << "Finished writing emissions." << std::endl;
        }

        void scheduleTasks(const std::vector<CtPointRadianceTask>& tasks)        // :176-192
        {
            taskRepeatCount = MAX_THREAD_COUNT / (uint32_t)tasks.size();
            if (taskRepeatCount == 0) throw std::runtime_error("more tasks than threads");
            threadsCount = (uint32_t)tasks.size() * taskRepeatCount;
            tasksBuffer.assign(threadsCount, CtPointRadianceTask{});
            for (uint32_t i = 0; i < tasks.size(); i++)
            {
                tasksBuffer[(size_t)i * taskRepeatCount] = tasks[i];            // slot 0 keeps the statistics gathered so far
                for (uint32_t j = 1; j < taskRepeatCount; j++)
                {
                    CtPointRadianceTask fresh{};
                    fresh.id = tasks[i].id;
                    std::memcpy(fresh.position, tasks[i].position, sizeof fresh.position);
                    std::memcpy(fresh.direction, tasks[i].direction, sizeof fresh.direction);
                    tasksBuffer[(size_t)i * taskRepeatCount + j] = fresh;
                }
            }
        }

        std::shared_ptr<Context> context;
        std::shared_ptr<Dataset> dataset;
        BatchSettings settings;
        std::vector<CtPointRadianceTask> tasksBuffer, convergedTasks;
        uint32_t taskRepeatCount = 0, threadsCount = 0, frameId = 0;
        bool allPixelsConverged = false;
    };

    class DisneyDescriptorCollector : public SceneItem                         // DisneyDescriptorCollector.cpp:15-100
    {
    public:
        DisneyDescriptorCollector(std::shared_ptr<Context> context, std::shared_ptr<Dataset> dataset, BatchSettings settings)
            : context(std::move(context)), dataset(std::move(dataset)), settings(settings) {}

        void init() override                                                    // :25-41: the batch's ScatterSample records
        {
            positions.resize(3 * (size_t)settings.batchSize);
            directions.resize(3 * (size_t)settings.batchSize);
            for (uint32_t i = 0; i < settings.batchSize; i++)
                Persistance::readScatterSample(dataset->getRecord("ScatterSample", (int32_t)(settings.batchStartId + i)), &positions[3 * (size_t)i],
                                               &directions[3 * (size_t)i]);
        }
        void update() override                                                  // collect() :57-63 + recordToDataset() :73-100
        {
            if (done) return;
            std::vector<uint8_t> grids((size_t)settings.batchSize * CT_DESCRIPTOR_BYTES);
            Context::check(ct_collect_descriptors(context->handle, positions.data(), directions.data(), settings.batchSize, grids.data()),
                           context->handle, "ct_collect_descriptors");
            std::vector<std::string> records(settings.batchSize);
            for (uint32_t i = 0; i < settings.batchSize; i++)
                records[i] = Persistance::disneyDescriptor(&grids[(size_t)i * CT_DESCRIPTOR_BYTES], CT_DESCRIPTOR_BYTES);
            dataset->batchAppend("DisneyDescriptor", records, (int32_t)settings.batchStartId);
            done = true;
        }
        bool isCompleted() override { return done; }

    private:
        std::shared_ptr<Context> context;
        std::shared_ptr<Dataset> dataset;
        BatchSettings settings;
        std::vector<float> positions, directions;
        bool done = false;
    };
}
